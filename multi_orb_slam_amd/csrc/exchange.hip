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
#include <memory>
#include <cstring>
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
    int (*CommGetAsyncError)(void*, int*) = nullptr;              // (optional)
    int (*CommAbort)(void*) = nullptr;                            // (optional)
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
            api.CommGetAsyncError = (int (*)(void*, int*))dlsym(api.lib, "ncclCommGetAsyncError");
            api.CommAbort = (int (*)(void*))dlsym(api.lib, "ncclCommAbort");
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

// A second, INDEPENDENT communicator over the same ranks (collective: every rank calls, in the same order).  ncclCommSplit where the
// library has it; otherwise (or when the split fails, or with MORB_EXCHANGE_ONE_COMM=split-off to force this path in tests) a fresh
// ncclCommInitRank whose id rank 0 draws and ships to the others with an all-gather over `comm` itself.  Never the same communicator
// twice (collectives of different steps sharing one communicator pair up by issue order, and a re-shipped block can be issued
// at another position on another rank): when no independent communicator can be made *out = nullptr and the caller decides
// (frontend.hip falls back to the arrangement that needs one communicator only, on every rank).
int morb::exchange_comm_clone(void* comm, int world, int rank, void** out) {
    RcclApi& R = rccl();
    *out = nullptr;
    const char* one = getenv("MORB_EXCHANGE_ONE_COMM");
    if (one && !strcmp(one, "1")) return ORB_OK;                         // (tests: "no second communicator can be made")
    const bool no_split = one && !strcmp(one, "split-off");
    if (R.CommSplit && !no_split) {
        void* c2 = nullptr;
        const int r = R.CommSplit(comm, /*color*/ 0, /*key*/ rank, &c2, nullptr);
        if (r == 0 && c2) { *out = c2; return ORB_OK; }
    }
    // a fresh id from rank 0, gathered over the parent (every rank contributes 128 bytes, all keep rank 0's)
    XUniqueId id; memset(&id, 0, sizeof id);
    if (rank == 0 && R.GetUniqueId(&id) != 0) memset(&id, 0, sizeof id);  // (an all-zero id makes every rank give up together)
    uint8_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)(world + 1) * 128) != hipSuccess) { (void)hipGetLastError(); return ORB_OK; }
    bool ok = hipMemcpy(d, id.internal, 128, hipMemcpyHostToDevice) == hipSuccess;
    hipStream_t ts = nullptr;   // (a stream of its own: the null stream would synchronise with every other stream of the process)
    ok = ok && hipStreamCreateWithFlags(&ts, hipStreamNonBlocking) == hipSuccess;
    ok = ok && R.AllGather(d, d + 128, 128, /*ncclUint8*/ 1, comm, ts) == 0 && hipStreamSynchronize(ts) == hipSuccess;
    if (ts) (void)hipStreamDestroy(ts);
    ok = ok && hipMemcpy(id.internal, d + 128, 128, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    bool zero = true;
    for (int i = 0; i < 128; ++i) zero = zero && id.internal[i] == 0;
    if (!ok || zero) { (void)hipGetLastError(); return ORB_OK; }
    void* c2 = nullptr;
    if (R.CommInitRank(&c2, world, id, rank) == 0 && c2) *out = c2;
    return ORB_OK;
}

// 0 = no asynchronous error on the communicator (or the library cannot say)
int morb::exchange_comm_async_error(void* comm) {
    RcclApi& R = rccl();
    int e = 0;
    if (!R.CommGetAsyncError || !comm) return 0;
    if (R.CommGetAsyncError(comm, &e) != 0) return -1;
    if (e) morb::set_error("RCCL reports an asynchronous error on the exchange's communicator: %s", R.GetErrorString ? R.GetErrorString(e) : "?");
    return e;
}

// after a timeout or an asynchronous error: a destroy would wait for the collective that never completes
void morb::exchange_comm_abort(void* comm) {
    if (!comm) return;
    RcclApi& R = rccl();
    if (R.CommAbort) (void)R.CommAbort(comm); else (void)R.CommDestroy(comm);
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


// ---- peer transport (round 6): the all-gather as DIRECT ONE-HOP WRITES into every rank's receive arena (every
// peer of an MI355X node is one xGMI hop away), between PROCESSES -- one per GPU, or several per GPU on a machine with fewer GPUs
// than ranks, which RCCL refuses.  Every rank allocates one fine-grained arena (blocks of all ranks for every exchange slot, first
// shipment and re-shipment apart, plus one arrival word per (slot, shipment, source rank)), exports it with hipIpcGetMemHandle and
// opens the others' (the caller carries the 64-byte handles between the processes: torch.distributed over gloo, MPI, a file).  An
// exchange is then ONE kernel that copies this rank's block into every arena and -- its last workgroup, behind a system-scope
// release -- stores the step's version number into this rank's arrival word there, and ONE single-wave kernel that waits until
// every source's word in the local arena carries the version (bounded: after MORB_EXCHANGE_TIMEOUT_MS it records which ranks are
// missing in mapped host memory and ends, so a dead peer is an error return of the step, never a hang).  No collective, no
// ordering between the exchanges of different steps, no communicator: a re-shipped block is just another version in another region.
namespace morb {
struct PeerComm {
    int world = 0, rank = 0, nslots = 0;
    size_t block = 0, flags_off = 0, bytes = 0;
    uint8_t* local = nullptr;
    std::vector<uint8_t*> peer;            // peer[r]: rank r's arena in this process' address space (peer[rank] == local)
    DevBuf<uint8_t*> d_peer;
    DevBuf<unsigned> d_counter;            // one per (slot, shipment): workgroups of k_peer_put that have finished
    PinnedBuf<uint32_t> missing;           // [nslots][2][2]: ranks whose block had not arrived when the wait gave up (64-bit mask)
    unsigned long long timeout_ticks = 0;  // wall_clock64 ticks (100 MHz)
    bool opened = false;
};
}  // namespace morb
using morb::PeerComm;

namespace {
constexpr int PEER_CHUNK = 16384;   // bytes per workgroup of k_peer_put (256 lanes x 4 x 16 B)

__global__ __launch_bounds__(256) void k_peer_put(const uint8_t* __restrict__ send, size_t block, uint8_t* const* __restrict__ peer,
                                                  size_t dst_off, size_t flag_off, int world, int chunks, unsigned version,
                                                  unsigned* __restrict__ counter) {
    const int p = blockIdx.x / chunks, c = blockIdx.x % chunks;
    uint8_t* dst = peer[p] + dst_off;
    const size_t b0 = (size_t)c * PEER_CHUNK, b1 = min(block, b0 + PEER_CHUNK);
    for (size_t o = b0 + (size_t)threadIdx.x * 16; o < b1; o += 256 * 16)
        *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(send + o);
    __threadfence_system();
    __syncthreads();
    __shared__ unsigned s_last;
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    // every workgroup's stores are released: the arrival words say so
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < world)
        __hip_atomic_store(reinterpret_cast<unsigned*>(peer[threadIdx.x] + flag_off), version, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_peer_wait(const unsigned* __restrict__ flags, int world, unsigned version,
                                                  unsigned long long timeout_ticks, uint32_t* __restrict__ missing2) {
    const int r = threadIdx.x;
    bool late = false;
    if (r < world) {
        const unsigned long long t0 = wall_clock64();
        // (versions only grow: a word already ahead of `version` belongs to a later step of a rank that cannot have got there without
        //  this one -- see frontend.hip on the slots -- and is refused as an error by the host through the block's own step number)
        while ((int)(__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - version) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > timeout_ticks) { late = true; break; }
        }
    }
    const unsigned long long m = __ballot(late);
    if (r == 0) {
        missing2[0] = (uint32_t)m; missing2[1] = (uint32_t)(m >> 32);
        __threadfence_system();
    }
}
}  // namespace

size_t morb::peer_handle_bytes() { return sizeof(hipIpcMemHandle_t); }
int morb::peer_world(const PeerComm* C) { return C->world; }
int morb::peer_rank(const PeerComm* C) { return C->rank; }

// this rank's arena + its handle for the other ranks (block: bytes of one export block; nslots: exchanges in flight)
int morb::peer_export(PeerComm** out, int world, int rank, size_t block, int nslots, long timeout_ms, uint8_t* handle) {
    MORB_ARG(out && handle && world >= 1 && world <= 64 && rank >= 0 && rank < world && block % 16 == 0 && nslots >= 1);
    std::unique_ptr<PeerComm> C(new PeerComm());
    C->world = world; C->rank = rank; C->nslots = nslots; C->block = block;
    C->flags_off = (((size_t)nslots * 2 * world * block) + 255) & ~(size_t)255;
    C->bytes = C->flags_off + (size_t)nslots * 2 * world * sizeof(unsigned);
    C->bytes = (C->bytes + 4095) & ~(size_t)4095;
    // fine-grained: remote writes must be visible to a kernel that is already running here (the wait), and nothing of it may linger
    // in this device's L2 when the next kernel reads the blocks
    if (hipExtMallocWithFlags((void**)&C->local, C->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        MORB_HIP(hipMalloc((void**)&C->local, C->bytes));
    }
    {   // zeroed on a stream of its own: no device-wide synchronisation (another thread's front end may be capturing its launch chain)
        hipStream_t zs = nullptr;
        MORB_HIP(hipStreamCreateWithFlags(&zs, hipStreamNonBlocking));
        hipError_t e1 = hipMemsetAsync(C->local, 0, C->bytes, zs), e2 = hipStreamSynchronize(zs);
        (void)hipStreamDestroy(zs);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(C->local); morb::set_error("peer arena: hipMemsetAsync failed"); return ORB_E_HIP; }
    }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, C->local) != hipSuccess) {
        morb::set_error("hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed where the driver only has dmabuf IPC)", hipGetErrorString(hipGetLastError()));
        (void)hipFree(C->local);
        return ORB_E_HIP;
    }
    memcpy(handle, &h, sizeof h);
    int rc;
    if ((rc = C->d_peer.reserve(world)) || (rc = C->d_counter.reserve((size_t)nslots * 2)) || (rc = C->missing.reserve((size_t)nslots * 4))) { (void)hipFree(C->local); return rc; }
    MORB_HIP(hipMemset(C->d_counter.p, 0, (size_t)nslots * 2 * sizeof(unsigned)));   // (synchronous for the host: a few bytes)
    memset(C->missing.p, 0, (size_t)nslots * 4 * sizeof(uint32_t));
    C->timeout_ticks = (unsigned long long)std::max(1L, timeout_ms) * 100000ull;   // (wall_clock64: 100 MHz)
    C->peer.assign(world, nullptr);
    C->peer[rank] = C->local;
    *out = C.release();
    return ORB_OK;
}

// handles: world x peer_handle_bytes(), rank r's at r (this rank's own entry is ignored)
int morb::peer_open(PeerComm* C, const uint8_t* handles) {
    MORB_ARG(C && handles && !C->opened);
    for (int r = 0; r < C->world; ++r) {
        if (r == C->rank) continue;
        hipIpcMemHandle_t h; memcpy(&h, handles + (size_t)r * sizeof h, sizeof h);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            morb::set_error("hipIpcOpenMemHandle (arena of rank %d) failed: %s", r, hipGetErrorString(hipGetLastError()));
            for (int q = 0; q < r; ++q) if (q != C->rank && C->peer[q]) { (void)hipIpcCloseMemHandle(C->peer[q]); C->peer[q] = nullptr; }
            return ORB_E_HIP;
        }
        C->peer[r] = static_cast<uint8_t*>(p);
    }
    MORB_HIP(hipMemcpy(C->d_peer.p, C->peer.data(), (size_t)C->world * sizeof(uint8_t*), hipMemcpyHostToDevice));
    C->opened = true;
    return ORB_OK;
}

void morb::peer_close(PeerComm* C) {
    if (!C) return;
    for (int r = 0; r < C->world; ++r) if (r != C->rank && C->peer[r]) (void)hipIpcCloseMemHandle(C->peer[r]);
    if (C->local) (void)hipFree(C->local);
    C->d_peer.release(); C->d_counter.release(); C->missing.release();
    delete C;
}

// the gathered blocks of (slot, shipment) in the local arena, rank-major
const uint8_t* morb::peer_recv(const PeerComm* C, int slot, int gen) { return C->local + ((size_t)(slot * 2 + gen) * C->world) * C->block; }

// one exchange on `st`: put this rank's block everywhere, wait for everybody's.  version: the step number + 1 (grows per slot).
int morb::peer_allgather(PeerComm* C, int slot, int gen, unsigned version, const void* send, hipStream_t st) {
    MORB_ARG(C && C->opened && slot >= 0 && slot < C->nslots && (gen == 0 || gen == 1) && send);
    const int chunks = (int)((C->block + PEER_CHUNK - 1) / PEER_CHUNK);
    const size_t region = (size_t)(slot * 2 + gen) * C->world;
    const size_t dst_off = (region + C->rank) * C->block;
    const size_t flag_off = C->flags_off + (region + C->rank) * sizeof(unsigned);
    uint32_t* missing = C->missing.dp + (size_t)(slot * 2 + gen) * 2;
    hipLaunchKernelGGL(k_peer_put, dim3(C->world * chunks), dim3(256), 0, st, static_cast<const uint8_t*>(send), C->block,
                       (uint8_t* const*)C->d_peer.p, dst_off, flag_off, C->world, chunks, version, C->d_counter.p + (slot * 2 + gen));
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned*>(C->local + C->flags_off + region * sizeof(unsigned)),
                       C->world, version, C->timeout_ticks, missing);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

// after the exchange's event: the ranks that had not delivered when the wait gave up (0: everybody had)
unsigned long long morb::peer_missing(const PeerComm* C, int slot, int gen) {
    const volatile uint32_t* m = C->missing.p + (size_t)(slot * 2 + gen) * 2;
    return (unsigned long long)m[0] | ((unsigned long long)m[1] << 32);
}
