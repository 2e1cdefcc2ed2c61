// exchange.hip -- the one collective of a multi-GPU timestep: an all-gather of every rank's descriptor export block.
//   * RCCL's C API resolved at run time (the copy torch.distributed already loaded, else the ROCm one): no link-time
//     dependency; a front end holds a few communicators over the same ranks (its steps' exchanges go round them by step
//     number, each issued at the tail of the step's extraction chain on that chain's stream: frontend.hip);
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
#include <map>
#include "loop_rendezvous.h"

// ---- loopback transport: the same exchange between front ends of ONE process on ONE device (one host thread per "rank").
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so a 1-GPU machine could otherwise never run the world > 1
// path of orbf_step.  The all-gather becomes: every rank announces its block and an event behind the work that produced it,
// all ranks rendezvous on the host, then every rank copies every block into its own receive buffer on ITS stream, behind the
// producers' events.  Same contract as the collective (every rank calls once per step, same block size); everything
// downstream -- k_repack_gathered, the rig-wide top-2, the early / late placement of the exchange inside a step -- is the
// product code unchanged.
namespace morb {
struct LoopGroup {
    Rendezvous rv;                   // (loop_rendezvous.h: the host-side meeting of the members, HIP-free and TSan-tested)
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
    int (*CommSplit)(void*, int, int, void**, void*) = nullptr;   // (optional: RCCL >= 2.18)
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
            api.CommSplit = (int (*)(void*, int, int, void**, void*))dlsym(api.lib, "ncclCommSplit");
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

// A second communicator over the same ranks (collective: every rank calls, in the same order).  *out = comm itself where the library
// has no ncclCommSplit or the split fails: the caller's collectives then share one communicator, which RCCL serialises in issue order.
int morb::exchange_comm_clone(void* comm, int rank, void** out) {
    RcclApi& R = rccl();
    *out = comm;
    if (!R.CommSplit || getenv("MORB_EXCHANGE_ONE_COMM")) return ORB_OK;
    void* c2 = nullptr;
    const int r = R.CommSplit(comm, /*color*/ 0, /*key*/ rank, &c2, nullptr);
    if (r == 0 && c2) *out = c2;
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
        G->rv.world = world; G->send.assign(world, nullptr); G->ev.assign(world, nullptr);
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
        bool full;
        { std::lock_guard<std::mutex> lk2(G->rv.mu); full = G->rv.members >= world; }
        if (G->rv.world != world || full) { morb::set_error("loopback group %d: world size mismatch or group full", group); return ORB_E_ARG; }
    }
    G->rv.join();
    *out = new LoopComm{G, rank};
    return ORB_OK;
}

// a member leaves: the group cannot exchange any more; its last member frees it
void morb::loop_leave(LoopComm* C) {
    if (!C) return;
    LoopGroup* G = C->g;
    if (G->rv.leave()) {
        std::lock_guard<std::mutex> lk(g_loop_mu);
        for (auto it = g_loop_groups.begin(); it != g_loop_groups.end(); ++it) if (it->second == G) { g_loop_groups.erase(it); break; }
        for (hipEvent_t e : G->ev) if (e) (void)hipEventDestroy(e);
        delete G;
    }
    delete C;
}

static int loop_fail(morb::Rendezvous::Result r, int rank) {
    using R = morb::Rendezvous;
    if (r == R::BROKEN_BEFORE) morb::set_error("loopback exchange: a member has left the group");
    else if (r == R::MEMBER_LEFT) morb::set_error("loopback exchange: a member left the group while rank %d waited for it", rank);
    else morb::set_error("loopback exchange: rank %d waited 20 s for the other ranks of its group", rank);
    return r == R::BROKEN_BEFORE ? ORB_E_ARG : ORB_E_HIP;
}

// Every member calls once per round, in the same order on every member (the host blocks until all have: a member that issues its
// rounds in another order than its peers -- e.g. one that stopped extracting ahead after a quadtree fallback while a block has to be
// shipped a second time -- can wait for peers that wait for it; RCCL calls return at once and have no such restriction).
int morb::loop_allgather(LoopComm* C, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st) {
    LoopGroup& G = *C->g;
    bool rec_ok = true;
    Rendezvous::Result r = G.rv.arrive([&] {
        G.send[C->rank] = sendbuf;
        rec_ok = hipEventRecord(G.ev[C->rank], st) == hipSuccess;
    });
    if (!rec_ok) { morb::set_error("loopback exchange: hipEventRecord failed"); return ORB_E_HIP; }
    if (r != Rendezvous::OK) return loop_fail(r, C->rank);
    for (int s = 0; s < G.rv.world; ++s) {
        MORB_HIP(hipStreamWaitEvent(st, G.ev[s], 0));
        MORB_HIP(hipMemcpyAsync((uint8_t*)recvbuf + (size_t)s * bytes, G.send[s], bytes, hipMemcpyDeviceToDevice, st));
    }
    // nobody re-records its event / republishes its block before every rank has enqueued this round's copies
    r = G.rv.arrive();
    if (r != Rendezvous::OK) return loop_fail(r, C->rank);
    return ORB_OK;
}
