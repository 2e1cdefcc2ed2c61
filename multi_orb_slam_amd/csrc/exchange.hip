// exchange.hip -- the one collective of a multi-GPU timestep: an all-gather of every rank's descriptor export block.
//   * RCCL's C API resolved at run time (the copy torch.distributed already loaded, else the ROCm one): no link-time
//     dependency, one communicator per front end, the collective issued on the matcher's side stream from inside the step;
//   * an in-process loopback transport with the same contract for machines with fewer GPUs than ranks.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/orbm.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "matcher_internal.h"
#include <dlfcn.h>
#include <condition_variable>
#include <map>

// ---- loopback transport: the same exchange between front ends of ONE process on ONE device (one host thread per "rank").
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so a 1-GPU machine could otherwise never run the world > 1
// path of orbf_step.  The all-gather becomes: every rank announces its block and an event behind the work that produced it,
// all ranks rendezvous on the host, then every rank copies every block into its own receive buffer on ITS stream, behind the
// producers' events.  Same contract as the collective (every rank calls once per step, same block size); everything
// downstream -- k_repack_gathered, the rig-wide top-2, the early / late placement of the exchange inside a step -- is the
// product code unchanged.
namespace morb {
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0, members = 0;
    unsigned long generation = 0;
    bool broken = false;
    std::vector<const void*> send;
    std::vector<hipEvent_t> ev;
};
struct LoopComm { LoopGroup* g; int rank; };
}  // namespace morb
using morb::LoopComm;
using morb::LoopGroup;

namespace {
struct XUniqueId { char internal[128]; };   // ncclUniqueId (rccl.h:43)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(XUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, XUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};
RcclApi& rccl() {
    static RcclApi api;
    if (!api.lib) {
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (api.lib) break;
        }
        if (!api.lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (api.lib) {
            api.GetUniqueId = (int (*)(XUniqueId*))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void**, int, XUniqueId, int))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
            api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
            api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        }
    }
    return api;
}
int rccl_fail(const char* what, int r) {
    RcclApi& R = rccl();
    morb::set_error("%s failed: %s", what, R.GetErrorString ? R.GetErrorString(r) : "RCCL error");
    return ORB_E_HIP;
}

std::mutex g_loop_mu;
std::map<int, LoopGroup*> g_loop_groups;

}  // namespace

int morb::exchange_rccl_available() { return rccl().ok() ? 1 : 0; }

int morb::exchange_unique_id(uint8_t* out128) {
    RcclApi& R = rccl();
    if (!R.ok()) { morb::set_error("librccl is not available"); return ORB_E_HIP; }
    XUniqueId id;
    const int r = R.GetUniqueId(&id);
    if (r) return rccl_fail("ncclGetUniqueId", r);
    memcpy(out128, id.internal, 128);
    return ORB_OK;
}

int morb::exchange_comm_init(void** comm, int world, const uint8_t* uid128, int rank) {
    RcclApi& R = rccl();
    if (!R.ok()) { morb::set_error("librccl is not available"); return ORB_E_HIP; }
    XUniqueId id; memcpy(id.internal, uid128, 128);
    const int r = R.CommInitRank(comm, world, id, rank);
    if (r) return rccl_fail("ncclCommInitRank", r);
    return ORB_OK;
}

void morb::exchange_comm_destroy(void* comm) { if (comm) (void)rccl().CommDestroy(comm); }

int morb::exchange_allgather(void* comm, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st) {
    const int r = rccl().AllGather(sendbuf, recvbuf, bytes, /*ncclUint8*/ 1, comm, st);
    if (r) return rccl_fail("ncclAllGather", r);
    return ORB_OK;
}

// a member joins the loopback group `group` (created by its first member)
int morb::loop_join(int group, int world, int rank, LoopComm** out) {
    LoopGroup* G = nullptr;
    std::lock_guard<std::mutex> lk(g_loop_mu);
    auto it = g_loop_groups.find(group);
    if (it == g_loop_groups.end()) {
        G = new LoopGroup();
        G->world = world; G->send.assign(world, nullptr); G->ev.assign(world, nullptr);
        for (int r = 0; r < world; ++r)
            if (hipEventCreateWithFlags(&G->ev[r], hipEventDisableTiming) != hipSuccess) {
                for (hipEvent_t e : G->ev) if (e) (void)hipEventDestroy(e);
                delete G;
                morb::set_error("loopback exchange: hipEventCreate failed");
                return ORB_E_HIP;
            }
        g_loop_groups[group] = G;
    } else {
        G = it->second;
        if (G->world != world || G->members >= world) { morb::set_error("loopback group %d: world size mismatch or group full", group); return ORB_E_ARG; }
    }
    {
        std::lock_guard<std::mutex> lk2(G->mu);
        ++G->members;
    }
    *out = new LoopComm{G, rank};
    return ORB_OK;
}

// a member leaves: the group cannot exchange any more; its last member frees it
void morb::loop_leave(LoopComm* C) {
    if (!C) return;
    LoopGroup* G = C->g;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(G->mu);
        G->broken = true; G->cv.notify_all();
        last = --G->members == 0;
    }
    if (last) {
        std::lock_guard<std::mutex> lk(g_loop_mu);
        for (auto it = g_loop_groups.begin(); it != g_loop_groups.end(); ++it) if (it->second == G) { g_loop_groups.erase(it); break; }
        for (hipEvent_t e : G->ev) if (e) (void)hipEventDestroy(e);
        delete G;
    }
    delete C;
}

int morb::loop_allgather(LoopComm* C, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st) {
    LoopGroup& G = *C->g;
    {
        std::unique_lock<std::mutex> lk(G.mu);
        if (G.broken) { morb::set_error("loopback exchange: a member has left the group"); return ORB_E_ARG; }
        G.send[C->rank] = sendbuf;
        if (hipEventRecord(G.ev[C->rank], st) != hipSuccess) { morb::set_error("loopback exchange: hipEventRecord failed"); return ORB_E_HIP; }
        const unsigned long gen = G.generation;
        if (++G.arrived == G.world) { G.arrived = 0; ++G.generation; G.cv.notify_all(); }
        else {
            // (the round is complete when the generation has moved on -- a member that leaves right BEHIND a completed round sets
            // `broken` before a slower waiter of that round has woken up, which must not fail the round it has just finished:
            // one such false alarm per ~1000 rig runs before round 4's soak found it)
            (void)G.cv.wait_for(lk, std::chrono::seconds(20), [&] { return G.generation != gen || G.broken; });
            if (G.generation == gen) {
                const bool left = G.broken;
                G.broken = true; G.cv.notify_all();
                if (left) morb::set_error("loopback exchange: a member left the group while rank %d waited for it", C->rank);
                else morb::set_error("loopback exchange: rank %d waited 20 s for the other ranks of its group", C->rank);
                return ORB_E_HIP;
            }
        }
    }
    for (int s = 0; s < G.world; ++s) {
        MORB_HIP(hipStreamWaitEvent(st, G.ev[s], 0));
        MORB_HIP(hipMemcpyAsync((uint8_t*)recvbuf + (size_t)s * bytes, G.send[s], bytes, hipMemcpyDeviceToDevice, st));
    }
    {   // nobody re-records its event / republishes its block before every rank has enqueued this round's copies
        std::unique_lock<std::mutex> lk(G.mu);
        const unsigned long gen = G.generation;
        if (++G.arrived == G.world) { G.arrived = 0; ++G.generation; G.cv.notify_all(); }
        else {
            (void)G.cv.wait_for(lk, std::chrono::seconds(20), [&] { return G.generation != gen || G.broken; });
            if (G.generation == gen) {
                const bool left = G.broken;
                G.broken = true; G.cv.notify_all();
                if (left) morb::set_error("loopback exchange: a member left the group while rank %d waited for it", C->rank);
                else morb::set_error("loopback exchange: rank %d waited 20 s for the other ranks of its group", C->rank);
                return ORB_E_HIP;
            }
        }
    }
    return ORB_OK;
}
