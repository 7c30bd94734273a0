// common.hpp -- error/log plumbing shared by the host side of libhtool_mi355x.so
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace hm {

typedef std::complex<double> cplx;

// log levels in the order of the reference's writer (src/htool/misc/logger.hpp:17-32)
enum LogLevel { LOG_CRITICAL = 0, LOG_ERROR = 1, LOG_WARNING = 2, LOG_DEBUG = 3, LOG_INFO = 4 };
void log_message(int level, const std::string &msg);
void set_log_sink(void (*sink)(int, const char *));

struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};

#define HM_CHECK(cond, msg)                                                                   \
    do {                                                                                      \
        if (!(cond)) throw hm::Error(std::string(msg));                                       \
    } while (0)

std::string strprintf(const char *fmt, ...);

double wall_seconds();

} // namespace hm
