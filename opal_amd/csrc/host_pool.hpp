// host_pool.hpp -- a small persistent pool of host threads for the boundary work of opalSearchDatabase
// (reference src/opal.cpp:1435-1519 does all of it on one thread; here the caller hands over N separate
// host allocations and N result structs per call, and walking them is what an MI355X call is made of
// once the DP itself takes under two milliseconds).
//
// One job at a time (submissions from several caller threads queue up). The submitting thread works too
// (parallel_for) or goes on with its own work while the workers run the job (begin / end: opalSearchDatabase
// launches the search of the sequences that have reached the device while the workers are still gathering the rest).
// Tasks are handed out in increasing order.
//
// A job costs what its tasks cost: only as many sleeping workers are woken as there are tasks, the submitter waits for
// the TASKS (a counter), not for every worker to check in and out (with 63 workers that alone took 0.15 ms per job, on a
// call whose whole budget is 2 ms), and the first few workers keep polling for ~100 us after a job, so that the five
// jobs of one opalSearchDatabase call find them awake.
//
// The pool is created on first use and intentionally never destroyed: joining threads from a
// static destructor, after the HIP runtime has begun to shut down, is a known way to hang at exit.
// After fork() the child has no worker threads (only the forking thread survives): the pool notices the new pid and
// runs jobs inline there instead of waiting for workers that do not exist.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#include "options.hpp"

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
            if (opal_host::opts().hostThreads >= 1) nt = (unsigned)opal_host::opts().hostThreads;
            return (int)nt <= default_threads() ? &get() : new HostPool((int)nt);
        }();
        return *pool;
    }
    // number of threads a job can run on (workers + the submitting thread)
    int width() const { return nworkers_ + 1; }

    // fn(task) for task in [0, ntasks); returns when every task has finished.
    void parallel_for(int ntasks, const std::function<void(int)>& fn) {
        if (ntasks <= 0) return;
        if (ntasks == 1 || nworkers_ == 0 || getpid() != pid_) {
            for (int t = 0; t < ntasks; t++) fn(t);
            return;
        }
        std::lock_guard<std::mutex> submit(submit_);
        const uint32_t gen = publish(ntasks, fn, ntasks - 1);
        run_tasks(gen, fn, ntasks);
        wait_done(ntasks);
        retire();
    }

    // The same job run by the WORKERS ONLY: begin() returns at once, the caller does something else and calls end()
    // (exactly once) to wait for the last task. `fn` must stay alive until end() returns. The caller must not use the
    // pool in between. With no workers (one hardware thread, or a forked child) the tasks run inside begin().
    void begin(int ntasks, const std::function<void(int)>& fn) {
        submit_.lock();  // held until end(): one job at a time
        asyncTasks_ = 0;
        if (ntasks <= 0 || nworkers_ == 0 || getpid() != pid_) {
            for (int t = 0; t < ntasks; t++) fn(t);
            return;
        }
        publish(ntasks, fn, ntasks);
        asyncTasks_ = ntasks;
    }
    void end() {
        if (asyncTasks_ > 0) {
            wait_done(asyncTasks_);
            retire();
            asyncTasks_ = 0;
        }
        submit_.unlock();
    }

    // CPU cores this process may use on average (cgroup v2 cpu.max, v1 cfs quota), 0 = no limit known.
    static double cpu_quota_cores() {
        double cores = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            double period = 0;
            if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) cores = atof(q) / period;
            fclose(f);
            return cores;
        }
        double quota = 0, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lf", &quota) != 1) quota = 0; fclose(f); }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lf", &period) != 1) period = 0; fclose(f); }
        return quota > 0 && period > 0 ? quota / period : 0;
    }
    static int default_threads() {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        if (nt > 64) nt = 64;  // memcpy-bound work: more threads than memory channels do not help
        // A container with a CPU quota: one and a half times the quota, not every hardware thread. Measured on the GPU box (256
        // hardware threads, quota 16 cores), cold opalSearchDatabase call on C2, best of 12, two rounds: 64 threads 2.46-2.60 ms,
        // 48 threads 2.43-2.45, 32 threads 2.33-2.48, 24 threads 2.31-2.33, 16 threads 2.37-2.52 (profiles/r05_e2e_variants.txt) --
        // 64 runnable threads on 16 cores' worth of time get throttled in the middle of a gather.
        double quota = cpu_quota_cores();
        // One process per GPU on one node (torchrun, bench.py --gpus N): the ranks share the quota. LOCAL_WORLD_SIZE is the
        // launchers' name for how many they are.
        if (const char* lw = getenv("LOCAL_WORLD_SIZE")) {
            const int ranks = atoi(lw);
            if (ranks > 1) quota = (quota > 0 ? quota : (double)nt) / ranks;
        }
        if (quota > 0) {
            const unsigned cap = (unsigned)(1.5 * quota + 0.5);
            if (nt > cap) nt = cap < 4 ? 4 : cap;
        }
        if (opal_host::opts().hostThreads >= 1) nt = (unsigned)opal_host::opts().hostThreads;  // (read when the pool is created: once per process)
        return (int)nt;
    }

private:
    static constexpr int SPINNERS = 15;        // workers that keep polling for a while after a job
    static constexpr int SPIN_MICROSECONDS = 120;

    explicit HostPool(int threads) : nworkers_(threads > 1 ? threads - 1 : 0), pid_(getpid()) {
        for (int i = 0; i < nworkers_; i++) std::thread([this, i] { worker(i); }).detach();
    }
    // Make a job visible: the ticket carries the job's generation in its high half, so a worker that still holds an
    // older job can never claim a task of this one. Wakes at most `helpers` sleeping workers.
    uint32_t publish(int ntasks, const std::function<void(int)>& fn, int helpers) {
        uint32_t gen;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            ntasks_ = ntasks;
            gen = ++gen_;
            done_.store(0, std::memory_order_relaxed);
            ticket_.store((uint64_t)gen << 32, std::memory_order_release);
            genSeen_.store(gen, std::memory_order_release);
        }
        const int wake = helpers < nworkers_ ? helpers : nworkers_;
        if (wake >= nworkers_) cvWork_.notify_all();
        else for (int i = 0; i < wake; i++) cvWork_.notify_one();
        return gen;
    }
    void retire() {
        std::lock_guard<std::mutex> lk(m_);
        job_ = nullptr;
    }
    void run_tasks(uint32_t gen, const std::function<void(int)>& fn, int ntasks) {
        for (;;) {
            uint64_t cur = ticket_.load(std::memory_order_acquire);
            for (;;) {
                if ((uint32_t)(cur >> 32) != gen || (int)(uint32_t)cur >= ntasks) return;
                if (ticket_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) break;
            }
            fn((int)(uint32_t)cur);
            if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == ntasks) {
                std::lock_guard<std::mutex> lk(m_);
                cvDone_.notify_all();
            }
        }
    }
    void wait_done(int ntasks) {
        for (int spin = 0; spin < 4000; spin++) {
            if (done_.load(std::memory_order_acquire) >= ntasks) return;
            __builtin_ia32_pause();
        }
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [&] { return done_.load(std::memory_order_acquire) >= ntasks; });
    }
    void worker(int index) {
        uint32_t seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            int n;
            uint32_t gen;
            {
                std::unique_lock<std::mutex> lk(m_);
                cvWork_.wait(lk, [&] { return gen_ != seen; });
                seen = gen = gen_;
                fn = job_;
                n = ntasks_;
            }
            for (;;) {
                if (fn) run_tasks(gen, *fn, n);
                if (index >= SPINNERS) break;
                // keep polling for the next job for a little while (the jobs of one API call follow each other closely)
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(SPIN_MICROSECONDS);
                bool found = false;
                while (std::chrono::steady_clock::now() < until) {
                    if (genSeen_.load(std::memory_order_acquire) != seen) { found = true; break; }
                    for (int i = 0; i < 64; i++) __builtin_ia32_pause();
                }
                if (!found) break;
                std::lock_guard<std::mutex> lk(m_);
                seen = gen = gen_;
                fn = job_;
                n = ntasks_;
            }
        }
    }

    const int nworkers_;
    const pid_t pid_;     // the process that owns the worker threads
    int asyncTasks_ = 0;  // tasks of the job between begin() and end(), protected by submit_
    std::mutex m_, submit_;
    std::condition_variable cvWork_, cvDone_;
    const std::function<void(int)>* job_ = nullptr;  // protected by m_, together with ntasks_ and gen_
    int ntasks_ = 0;
    uint32_t gen_ = 0;
    std::atomic<uint32_t> genSeen_{0};   // gen_ again, for the polling workers
    std::atomic<uint64_t> ticket_{0};    // (generation << 32) | next task
    std::atomic<int> done_{0};
};
