// host_pool.hpp -- a small persistent pool of host threads for the boundary work of opalSearchDatabase
// (reference src/opal.cpp:1435-1519 does all of it on one thread; here the caller hands over N separate
// host allocations and N result structs per call, and walking them is what an MI355X call is made of
// once the DP itself takes under two milliseconds).
//
// One job at a time (submissions from several caller threads queue up); the submitting thread works
// too (parallel_for) or goes on with its own work while the workers run the job (begin / end: opalSearchDatabase
// launches the search of the sequences that have reached the device while the workers are still gathering the rest).
// The pool is created on first use and intentionally never destroyed: joining threads from a
// static destructor, after the HIP runtime has begun to shut down, is a known way to hang at exit.
// After fork() the child has no worker threads (only the forking thread survives): the pool notices the new pid and
// runs jobs inline there instead of waiting for workers that do not exist.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

class HostPool {
public:
    static HostPool& get() {
        static HostPool* pool = new HostPool(default_threads());
        return *pool;
    }
    // Compute-bound host work (the alignment traceback) scales past the 64 threads that saturate memory copies:
    // a second pool over every hardware thread, created only if such work is ever asked for.
    static HostPool& get_wide() {
        static HostPool* pool = [] {
            unsigned nt = std::thread::hardware_concurrency();
            if (nt == 0) nt = 1;
            if (nt > 256) nt = 256;
            if (const char* e = getenv("OPAL_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) nt = (unsigned)v; }
            return (int)nt <= default_threads() ? &get() : new HostPool((int)nt);
        }();
        return *pool;
    }
    // number of threads a job can run on (workers + the submitting thread)
    int width() const { return (int)workers_.size() + 1; }

    // fn(task) for task in [0, ntasks); returns when every task has finished.
    void parallel_for(int ntasks, const std::function<void(int)>& fn) {
        if (ntasks <= 0) return;
        if (ntasks == 1 || workers_.empty() || getpid() != pid_) {
            for (int t = 0; t < ntasks; t++) fn(t);
            return;
        }
        std::lock_guard<std::mutex> submit(submit_);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            ntasks_ = ntasks;
            next_.store(0, std::memory_order_relaxed);
            active_ = (int)workers_.size();
            gen_++;
        }
        cvWork_.notify_all();
        run_tasks(fn, ntasks);
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [&] { return active_ == 0; });
        job_ = nullptr;
    }

    // The same job run by the WORKERS ONLY: begin() returns at once, the caller does something else and calls end()
    // (exactly once) to wait for the last task. Tasks are handed out in increasing order. `fn` must stay alive until
    // end() returns. With no workers (one hardware thread, or a forked child) the tasks run inside begin().
    void begin(int ntasks, const std::function<void(int)>& fn) {
        submit_.lock();  // held until end(): one job at a time
        if (ntasks <= 0 || workers_.empty() || getpid() != pid_) {
            for (int t = 0; t < ntasks; t++) fn(t);
            asyncActive_ = false;
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            ntasks_ = ntasks;
            next_.store(0, std::memory_order_relaxed);
            active_ = (int)workers_.size();
            gen_++;
        }
        asyncActive_ = true;
        cvWork_.notify_all();
    }
    void end() {
        if (asyncActive_) {
            std::unique_lock<std::mutex> lk(m_);
            cvDone_.wait(lk, [&] { return active_ == 0; });
            job_ = nullptr;
            asyncActive_ = false;
        }
        submit_.unlock();
    }

    static int default_threads() {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        if (nt > 64) nt = 64;  // memcpy-bound work: more threads than memory channels do not help
        if (const char* e = getenv("OPAL_HOST_THREADS")) {
            const int v = atoi(e);
            if (v >= 1) nt = (unsigned)v;
        }
        return (int)nt;
    }

private:
    explicit HostPool(int threads) : pid_(getpid()) {
        for (int i = 1; i < threads; i++) {
            workers_.emplace_back([this] { worker(); });
            workers_.back().detach();
        }
    }
    void run_tasks(const std::function<void(int)>& fn, int ntasks) {
        for (;;) {
            const int t = next_.fetch_add(1, std::memory_order_relaxed);
            if (t >= ntasks) break;
            fn(t);
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cvWork_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                fn = job_;
                n = ntasks_;
            }
            if (fn) run_tasks(*fn, n);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) cvDone_.notify_all();
            }
        }
    }

    std::vector<std::thread> workers_;
    const pid_t pid_;           // the process that owns the worker threads
    bool asyncActive_ = false;  // between begin() and end(), protected by submit_
    std::mutex m_, submit_;
    std::condition_variable cvWork_, cvDone_;
    const std::function<void(int)>* job_ = nullptr;
    int ntasks_ = 0;
    std::atomic<int> next_{0};
    int active_ = 0;
    uint64_t gen_ = 0;
};
