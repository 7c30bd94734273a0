// common.hpp -- error/log plumbing shared by the host side of libhtool_mi355x.so
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace hm {

typedef std::complex<double> cplx;

// log levels in the order of the reference's writer (src/htool/misc/logger.hpp:17-32)
enum LogLevel { LOG_CRITICAL = 0, LOG_ERROR = 1, LOG_WARNING = 2, LOG_DEBUG = 3, LOG_INFO = 4 };
void log_message(int level, const std::string &msg);
void set_log_sink(void (*sink)(int, const char *));
// Messages of a helper thread are held back and emitted by the thread that owns the call: the sink may need a lock only that
// thread can give up (the Python shim's sink takes the interpreter lock).
struct DeferredLog { std::vector<std::pair<int, std::string>> held; };
void defer_log_to(DeferredLog *d); // this thread's messages go to d from now on (nullptr: straight to the sink again)
void flush_deferred(DeferredLog &d);

struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};

#define HM_CHECK(cond, msg)                                                                   \
    do {                                                                                      \
        if (!(cond)) throw hm::Error(std::string(msg));                                       \
    } while (0)

std::string strprintf(const char *fmt, ...);

double wall_seconds();

// OpenMP loop for translation units that are not compiled with OpenMP (the .hip file): fn(i, ctx) for i in [0, n)
void parallel_for_index(long long n, void (*fn)(long long, void *), void *ctx);
template <typename F>
inline void parallel_for(long long n, F &&f) {
    parallel_for_index(n, [](long long i, void *c) { (*static_cast<typename std::remove_reference<F>::type *>(c))(i); }, (void *)&f);
}

} // namespace hm
