// ThreadSanitizer run of the device group's launcher-thread pool (csrc/psa_group_workers.hpp) on the CPU build:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread group_workers_tsan.cpp
// The pool has no HIP calls of its own (device selection is an injected callable), so the whole doorbell / done-word /
// sleep-wake protocol runs here: rounds back to back (the hot-spin path), rounds behind pauses longer than hot_us (every
// thread asleep on the condition variable: the wake-up path), pauses of about hot_us (threads on their way to sleep while
// the doorbell rings: the lost-wake-up window), a destroy while idle-hot, a destroy while asleep, a destroy right after
// the last round.  Every round checks that each of the N jobs ran exactly once and saw that round's data.
// argv: [rounds = 12000] [workers = 8] [hot_us = 200].  Exit 0 and no TSan report = clean.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "psa_group_workers.hpp"

static int run_pool(int rounds, int workers, double hot_us, int destroy_how, unsigned seed)
{
    std::vector<int> devices((size_t)workers);
    for (int i = 0; i < workers; ++i) devices[(size_t)i] = 100 + i;
    std::vector<int> started((size_t)workers, 0);        // written by on_start on thread i, read after the first round
    int failures = 0;
    {
        psa::GroupWorkers pool(devices, [&](int device) { started[(size_t)(device - 100)] = device; }, hot_us);
        std::vector<long long> out((size_t)workers, 0);   // plain memory: a job's results, read by the caller after run()
        long long payload = 0;                            // plain memory: the call's arguments, written before run()
        unsigned lcg = seed;
        for (int r = 0; r < rounds; ++r) {
            payload = 1000003LL * r + 7;
            auto job = [&](int i) { out[(size_t)i] = payload + i; };
            pool.run(job);
            for (int i = 0; i < workers; ++i)
                if (out[(size_t)i] != payload + i) ++failures;
            if (r == 0)
                for (int i = 1; i < workers; ++i)
                    if (started[(size_t)i] != 100 + i) ++failures;
            // the pauses between rounds: mostly none, sometimes around hot_us (racing the threads to sleep), sometimes
            // several times hot_us (everybody asleep)
            lcg = lcg * 1664525u + 1013904223u;
            const unsigned pick = (lcg >> 16) % 100u;
            if (pick < 3u) std::this_thread::sleep_for(std::chrono::microseconds((long long)(hot_us * 4.0)));
            else if (pick < 10u) std::this_thread::sleep_for(std::chrono::microseconds((long long)(hot_us * (0.6 + 0.1 * (pick - 3u)))));
        }
        if (destroy_how == 1) std::this_thread::sleep_for(std::chrono::microseconds((long long)(hot_us * 0.3)));   // idle, hot
        if (destroy_how == 2) std::this_thread::sleep_for(std::chrono::microseconds((long long)(hot_us * 6.0)));   // idle, asleep
    }                                                     // ~GroupWorkers: stop, wake, join
    return failures;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 12000;
    const int workers = argc > 2 ? std::atoi(argv[2]) : 8;
    const double hot_us = argc > 3 ? std::atof(argv[3]) : 200.0;
    int failures = 0;
    long long total = 0;
    for (int how = 0; how < 3; ++how) {                  // destroy right after a round / while hot / while asleep
        const int n = how == 0 ? rounds : rounds / 20 + 1;
        failures += run_pool(n, workers, hot_us, how, 12345u + (unsigned)how);
        total += n;
    }
    failures += run_pool(50, 1, hot_us, 2, 99u);          // a group of one device has no threads at all
    failures += run_pool(0, workers, hot_us, 2, 7u);      // created and destroyed without a call (threads asleep)
    failures += run_pool(0, workers, hot_us, 0, 7u);      // ... and before the threads have even started
    std::printf("group_workers: %lld rounds over %d workers, %d failures\n", total, workers, failures);
    return failures ? 1 : 0;
}
