// ThreadSanitizer driver of the loopback exchange's rendezvous (multi_orb_slam_amd/csrc/loop_rendezvous.h): `world` threads meet twice
// per round (the two barriers of loop_allgather), publish a value under the lock and read everybody's after the first barrier -- the
// access pattern of the send-pointer table --, for thousands of rounds; then the scenarios around members leaving: right behind a
// completed round (no waiter of that round may fail), in the middle of a round (the waiters must be told), before anybody arrives.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../multi_orb_slam_amd/csrc/loop_rendezvous.h"

using morb::Rendezvous;

static int steady_rounds(int world, int rounds) {
    Rendezvous rv; rv.world = world;
    std::vector<long> slot(world, -1);
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) rv.join();
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            for (int k = 0; k < rounds; ++k) {
                if (rv.arrive([&] { slot[r] = (long)k * 1000 + r; }) != Rendezvous::OK) { ++bad; return; }
                for (int s = 0; s < world; ++s) if (slot[s] != (long)k * 1000 + s) ++bad;      // everybody's value of THIS round
                if (rv.arrive() != Rendezvous::OK) { ++bad; return; }                           // nobody republishes before all have read
            }
            rv.leave();
        });
    for (auto& t : th) t.join();
    return bad.load();
}

// every member leaves right behind its last round: a slower waiter of that round sees `broken` set and must still report OK
static int leave_behind_a_completed_round(int world, int reps) {
    int bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        Rendezvous* rv = new Rendezvous(); rv->world = world;
        std::atomic<int> fails{0}, last{0};
        for (int r = 0; r < world; ++r) rv->join();
        std::vector<std::thread> th;
        for (int r = 0; r < world; ++r)
            th.emplace_back([&] {
                for (int k = 0; k < 3; ++k) if (rv->arrive() != Rendezvous::OK) ++fails;
                if (rv->leave()) ++last;
            });
        for (auto& t : th) t.join();
        bad += fails.load() + (last.load() != 1);
        delete rv;
    }
    return bad;
}

// one member never arrives and leaves instead: the waiters are released with MEMBER_LEFT, later arrivals get BROKEN_BEFORE
static int leave_in_the_middle(int world) {
    Rendezvous rv; rv.world = world;
    for (int r = 0; r < world; ++r) rv.join();
    std::atomic<int> left{0}, before{0}, other{0};
    std::vector<std::thread> th;
    for (int r = 0; r < world - 1; ++r)
        th.emplace_back([&] {
            const Rendezvous::Result a = rv.arrive([] {}, std::chrono::seconds(5));
            if (a == Rendezvous::MEMBER_LEFT) ++left; else if (a == Rendezvous::BROKEN_BEFORE) ++before; else ++other;
            rv.leave();
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    rv.leave();
    for (auto& t : th) t.join();
    return (other.load() != 0) + (left.load() + before.load() != world - 1) + (left.load() < 1);
}

static int timeout_is_reported(int world) {
    Rendezvous rv; rv.world = world;
    rv.join();
    return rv.arrive([] {}, std::chrono::milliseconds(30)) == Rendezvous::TIMEOUT ? 0 : 1;
}

int main() {
    int bad = 0;
    bad += steady_rounds(4, 4000);
    bad += steady_rounds(8, 1000);
    bad += steady_rounds(2, 8000);
    bad += leave_behind_a_completed_round(4, 400);
    bad += leave_in_the_middle(4);
    bad += timeout_is_reported(3);
    std::printf("tsan_rendezvous: %d failures\n", bad);
    return bad ? 1 : 0;
}
