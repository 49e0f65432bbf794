// psa_group_workers.hpp — the launcher threads of a device group.  Host-only, no HIP calls: what a thread does when it
// starts (select its device) is an injected callable, so the CPU build can run this file under ThreadSanitizer
// (tests/c/group_workers_tsan.cpp).
//
// A group call has per-device work on the host (select the device, build the launch arguments, enqueue: ~5 us each);
// issued from one thread the 8th shard of a 0.1 ms search started ~40 us after the first.  The group owns ONE
// PERSISTENT THREAD PER DEVICE beyond the first (the caller's thread serves device 0): a call rings a doorbell (a
// generation counter), every thread runs the call's job for its device, the caller waits for their "done" words.
// A thread keeps polling the doorbell for hot_us after its last job (back-to-back calls are picked up within a
// microsecond), then sleeps on a condition variable (an idle group costs nothing; the first call after a pause pays
// one wake-up).  Every thread runs on_start(device) once, when it starts.
//
// Memory order of the hand-off: the job pointer is written before the doorbell (gen_, seq_cst store) and read after a
// thread has seen the new generation (acquire load): published.  A thread's results are written before its done word
// (release store) and read by the caller after it has seen the word (acquire load).  Sleep / wake: a thread raises
// sleepers_ under the mutex BEFORE it looks at the doorbell again (the wait's predicate); the caller rings BEFORE it
// looks at sleepers_; both seq_cst: at least one of the two sees the other, so no wake-up is lost.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace psa {

class GroupWorkers {
public:
    // devices[i]: what thread i hands to on_start (device 0's entry is the caller's own business)
    GroupWorkers(const std::vector<int>& devices, std::function<void(int)> on_start, double hot_us = 500.0)
        : n_((int)devices.size()), hot_us_(hot_us), done_((size_t)n_), on_start_(std::move(on_start))
    {
        for (auto& d : done_) d.v.store(0u);
        for (int i = 1; i < n_; ++i) threads_.emplace_back([this, i, dev = devices[(size_t)i]] { loop(i, dev); });
    }
    ~GroupWorkers()
    {
        stop_.store(true);
        { std::lock_guard<std::mutex> lock(m_); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    GroupWorkers(const GroupWorkers&) = delete;
    GroupWorkers& operator=(const GroupWorkers&) = delete;

    int size() const { return n_; }

    // job(i) for every device i of the group: i = 0 on the calling thread, the others on their own; returns when all have.
    // One call at a time (the group's entry points are not re-entrant).
    template <typename Job>
    void run(Job&& job)
    {
        std::function<void(int)> fn = std::ref(job);
        job_ = &fn;
        const unsigned g = gen_.load() + 1u;
        gen_.store(g);                                   // (seq_cst, like the sleepers' count: see loop)
        if (sleepers_.load() > 0) {
            { std::lock_guard<std::mutex> lock(m_); }
            cv_.notify_all();
        }
        job(0);
        for (int i = 1; i < n_; ++i) {
            // (a device's share of a blocking search ends within microseconds of the caller's own; a batch's may take
            // milliseconds: after a short spin the caller gives its core up between looks)
            unsigned spins = 0;
            while (done_[(size_t)i].v.load(std::memory_order_acquire) != g) {
                if (++spins < 4096u) cpu_relax();
                else std::this_thread::yield();
            }
        }
        job_ = nullptr;
    }

private:
    struct alignas(64) Done { std::atomic<unsigned> v; };

    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    void loop(int i, int device)
    {
        if (on_start_) on_start_(device);
        unsigned seen = 0u;
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            unsigned g = seen;
            for (unsigned spins = 1;; ++spins) {
                g = gen_.load(std::memory_order_acquire);
                if (g != seen || stop_.load(std::memory_order_relaxed)) break;
                cpu_relax();
                if ((spins & 1023u) == 0u &&
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - idle_since).count() > hot_us_) {
                    // to sleep.  The count goes up before the doorbell is looked at again (under the lock), the caller
                    // rings before it looks at the count: one of the two sees the other (both seq_cst).
                    std::unique_lock<std::mutex> lock(m_);
                    sleepers_.fetch_add(1);
                    cv_.wait(lock, [&] { return gen_.load() != seen || stop_.load(); });
                    sleepers_.fetch_sub(1);
                }
            }
            if (g == seen) return;                       // stop
            seen = g;
            (*job_)(i);
            done_[(size_t)i].v.store(g, std::memory_order_release);
            idle_since = std::chrono::steady_clock::now();
        }
    }

    int n_;
    double hot_us_;
    std::vector<Done> done_;
    std::vector<std::thread> threads_;
    std::atomic<unsigned> gen_{0u};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false};
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void(int)> on_start_;
    const std::function<void(int)>* job_ = nullptr;
};

}  // namespace psa
