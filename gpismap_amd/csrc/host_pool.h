// A small persistent thread pool for the data-parallel parts of update()'s host side (pure per-point arithmetic between the
// ObsGP batches and the sequential tree replay).  parallel_for(n, f) runs f(begin, end) over disjoint ranges on the
// workers and the caller; it returns when all ranges are done.  Results do not depend on the number of threads: every item
// is computed by the same code from the same inputs, only by another thread.  GPIS_HOST_THREADS sets the size (default
// min(8, hardware threads / 2); 1 = everything on the calling thread).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace gpis {

class HostPool {
public:
    HostPool() {
        int n = 0;
        if (const char* e = getenv("GPIS_HOST_THREADS")) n = atoi(e);
        if (n <= 0) n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
        nthreads_ = std::max(1, n);
        for (int i = 1; i < nthreads_; ++i) workers_.emplace_back([this, i] { worker(i); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return nthreads_; }
    // f(begin, end): called once per non-empty range; ranges partition [0, n)
    void parallel_for(int n, const std::function<void(int, int)>& f, int min_per_thread = 256) {
        const int parts = std::max(1, std::min(nthreads_, n / std::max(1, min_per_thread)));
        if (parts <= 1) { if (n > 0) f(0, n); return; }
        run_parts(parts, [&](int p) {
            const int lo = (int)((long long)n * p / parts), hi = (int)((long long)n * (p + 1) / parts);
            if (hi > lo) f(lo, hi);
        });
    }
    // g(part) for part = 0 .. parts-1 (parts <= size()); part p always runs on the same thread (0 = the caller), so data a
    // part writes in one call and reads or rewrites in the next stays in that core's cache
    void run_parts(int parts, const std::function<void(int)>& g) {
        parts = std::max(1, std::min(parts, nthreads_));
        if (parts == 1) { g(0); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &g; parts_ = parts; pending_ = parts - 1; ++gen_;
        }
        cv_.notify_all();
        g(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run_part(int p) { (*fn_)(p); }
    void worker(int id) {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            if (id >= parts_) continue;
            lk.unlock();
            run_part(id);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    int nthreads_ = 1;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int parts_ = 1, pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

}  // namespace gpis
