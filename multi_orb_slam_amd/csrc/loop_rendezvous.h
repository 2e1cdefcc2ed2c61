// loop_rendezvous.h -- the host-side rendezvous of the in-process loopback exchange (exchange.hip), free of HIP so that it can be
// built and run under ThreadSanitizer on its own (tests/san/tsan_rendezvous.cpp, `make -C tests/san`).
//
// `world` members meet once per round: arrive() returns when the last one has arrived.  A round is complete when the generation has
// moved on -- a member that leaves right BEHIND a completed round sets `broken` before a slower waiter of that round has woken up,
// which must not fail the round that waiter has just finished (one false alarm per ~1000 rig runs before round 4's soak found it).
#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>

namespace morb {

struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0, members = 0;
    unsigned long generation = 0;
    bool broken = false;

    enum Result { OK = 0, BROKEN_BEFORE = 1, MEMBER_LEFT = 2, TIMEOUT = 3 };

    // Runs `publish` under the lock (what this member contributes to the round: nobody reads it before the round is complete), then
    // waits for the others.
    template <class Publish>
    Result arrive(Publish&& publish, std::chrono::milliseconds patience = std::chrono::seconds(20)) {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return BROKEN_BEFORE;
        publish();
        const unsigned long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return OK; }
#if defined(__SANITIZE_THREAD__)
        // (gcc 11's libtsan does not intercept pthread_cond_clockwait -- what a steady-clock wait compiles to -- and then believes the
        // mutex stays held across the wait; the system-clock form goes through pthread_cond_timedwait, which it knows)
        (void)cv.wait_until(lk, std::chrono::system_clock::now() + patience, [&] { return generation != gen || broken; });
#else
        (void)cv.wait_for(lk, patience, [&] { return generation != gen || broken; });
#endif
        if (generation != gen) return OK;
        const bool left = broken;
        broken = true; cv.notify_all();
        return left ? MEMBER_LEFT : TIMEOUT;
    }
    Result arrive() { return arrive([] {}); }

    void join() { std::lock_guard<std::mutex> lk(mu); ++members; }
    // -> true for the last member (who frees the group)
    bool leave() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true; cv.notify_all();
        return --members == 0;
    }
};

}  // namespace morb
