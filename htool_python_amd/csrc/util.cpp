// util.cpp -- logging sink, string formatting, wall clock
#include <chrono>
#include <cstdarg>
#include <vector>

#include "common.hpp"

namespace hm {

static void (*g_sink)(int, const char *) = nullptr;

void set_log_sink(void (*sink)(int, const char *)) { g_sink = sink; }

static thread_local DeferredLog *t_deferred = nullptr;
void defer_log_to(DeferredLog *d) { t_deferred = d; }

void log_message(int level, const std::string &msg) {
    if (t_deferred) { t_deferred->held.emplace_back(level, msg); return; }
    if (g_sink) g_sink(level, msg.c_str());
}

void flush_deferred(DeferredLog &d) {
    for (auto &m : d.held) log_message(m.first, m.second);
    d.held.clear();
}

std::string strprintf(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    va_list ap2;
    va_copy(ap2, ap);
    int n = vsnprintf(nullptr, 0, fmt, ap);
    va_end(ap);
    std::vector<char> buf((size_t)n + 1);
    vsnprintf(buf.data(), buf.size(), fmt, ap2);
    va_end(ap2);
    return std::string(buf.data(), (size_t)n);
}

void parallel_for_index(long long n, void (*fn)(long long, void *), void *ctx) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; i++) fn(i, ctx);
}

double wall_seconds() {
    using namespace std::chrono;
    return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

} // namespace hm
