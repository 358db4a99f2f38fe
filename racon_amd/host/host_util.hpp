// host_util.hpp -- small helpers shared by the host layer's translation units.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "fatal.hpp"

namespace racon {

inline double seconds_since(const std::chrono::time_point<std::chrono::steady_clock>& t) {
    return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t).count();
}

// fn(i) for i in [0, n) on `threads` host threads (dynamic distribution)
template <class F>
inline void parallel_for(uint64_t n, uint32_t threads, F fn) {
    threads = std::max<uint32_t>(1, std::min<uint64_t>(threads, n));
    if (threads == 1) { for (uint64_t i = 0; i < n; ++i) fn(i); return; }
    // an exception inside a worker (fatal() in library mode, nw_path's runtime_error, bad_alloc) must not escape its thread
    // (std::terminate): the first one is kept and rethrown on the caller's thread after the join
    std::atomic<uint64_t> next{0};
    std::exception_ptr first_error;
    std::mutex error_mutex;
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            FatalThrowsScope scope;
            try {
                for (uint64_t i; (i = next.fetch_add(1)) < n;) fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!first_error) first_error = std::current_exception();
                next.store(n);                                     // the other workers stop at their next item
            }
        });
    for (auto& t : pool) t.join();
    if (first_error) fatal_from(first_error);
}

}  // namespace racon
