// worker_pool.hpp -- the host threads SDFViewer::update samples a host-only SDF on (sdf_viewer_ingest.cpp).
#pragma once

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sdfviewer {

// Persistent workers.  A SESSION spans one update() call: begin() wakes the background threads (condition variable, once),
// run(n, fn) -- any number of times -- calls fn(0) on the calling thread and fn(1) .. fn(n - 1) on background threads and
// returns when all are done, end() parks them again.  Inside a session the workers SPIN on a generation counter between runs:
// a run lasts a fraction of a millisecond, and waking 63 threads through one mutex costs about as much (measured on a
// 2 x 64-core host: 64 workers reached 25 % of their single-thread rate with a condition variable per run).
// One caller at a time (begin / run / end are the owning viewer's, which is single-owner like everything behind the C ABI).
class WorkerPool {
   public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void begin(unsigned n) {
        while (threads_.size() + 1 < n) {
            const unsigned id = (unsigned)threads_.size() + 1;
            // the newcomer must take the NEXT run for news: it starts from the last generation published (only this thread
            // publishes), not from whatever it finds when it gets to look
            const unsigned long long last = generation_.load(std::memory_order_relaxed);
            threads_.emplace_back([this, id, last] { loop(id, last); });
        }
        if (n <= 1) return;
        {
            std::lock_guard<std::mutex> lock(m_);
            session_.store(true, std::memory_order_release);
        }
        wake_.notify_all();
    }
    void end() { session_.store(false, std::memory_order_release); }
    void run(unsigned n, const std::function<void(unsigned)>& fn) {
        if (n <= 1) {
            if (n == 1) fn(0);
            return;
        }
        fn_ = &fn;
        pending_.store(n - 1, std::memory_order_relaxed);
        // the run's number and its worker count travel in ONE word: a worker decides whether it takes part from the same
        // load that told it about the run (a bystander that is slow to look may skip runs, a participant cannot: the caller waits)
        sequence_ += 1;
        generation_.store(sequence_ << 16 | n, std::memory_order_release);
        fn(0);
        // the caller's own share may be much shorter than the workers' (it ships while they sample): spin briefly, then SLEEP --
        // a yielding spin still burns a CPU's worth of a cgroup quota the workers need
        for (unsigned spins = 0; spins < 256 && pending_.load(std::memory_order_acquire) != 0; ++spins) __builtin_ia32_pause();
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lock(done_m_);
            done_.wait(lock, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
    }
    // CPUs this process may actually use at once: the machine's hardware threads, cut down to the scheduler affinity mask and
    // to a cgroup CPU quota (cpu.max: a container limited to 16 CPUs' worth of time on a 256-thread host reports 256 hardware
    // threads; 64 workers there only take turns).
    static unsigned usable_cpus();
    static unsigned probe_usable_cpus();

   private:
    void loop(unsigned id, unsigned long long seen) {
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return stop_ || session_.load(std::memory_order_acquire); });
                if (stop_) return;
            }
            unsigned spins = 0;
            while (session_.load(std::memory_order_acquire)) {
                const unsigned long long g = generation_.load(std::memory_order_acquire);
                if (g == seen) {
                    relax(spins++);
                    continue;
                }
                spins = 0;
                seen = g;
                if (id < (g & 0xffff)) {  // (fn_ was written before the generation was published)
                    (*fn_)(id);
                    if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // the last one out wakes the caller
                        std::lock_guard<std::mutex> lock(done_m_);
                        done_.notify_one();
                    }
                }
            }
        }
    }
    // Waiting inside a session: spin for the first microseconds (the next run is normally that close), then give the core
    // away between looks -- with more workers than usable CPUs a pure spin starves the thread everybody waits for.
    static void relax(unsigned spins) {
        if (spins < 2048)
            __builtin_ia32_pause();
        else
            std::this_thread::yield();
    }
    std::mutex m_, done_m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned long long sequence_ = 0;
    std::atomic<unsigned> pending_{0};
    std::atomic<unsigned long long> generation_{0};
    std::atomic<bool> session_{false};
    bool stop_ = false;
};


inline unsigned WorkerPool::usable_cpus() {
    static const unsigned cached = [] { return probe_usable_cpus(); }();  // (asked once per process: update() runs every frame)
    return cached;
}

inline unsigned WorkerPool::probe_usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int k = CPU_COUNT(&set);
        if (k > 0 && (unsigned)k < n) n = (unsigned)k;
    }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char quota[32] = "";
        long long period = 0;
        if (fscanf(f, "%31s %lld", quota, &period) == 2 && period > 0 && quota[0] != 'm') {
            const long long q = atoll(quota);
            if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (q + period - 1) / period));
        }
        fclose(f);
    } else if (FILE* q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        long long quota = -1, period = 0;
        const bool ok = fscanf(q, "%lld", &quota) == 1;
        fclose(q);
        if (FILE* p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(p, "%lld", &period) != 1) period = 0;
            fclose(p);
        }
        if (ok && quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    }
    return n;
}

}  // namespace sdfviewer
