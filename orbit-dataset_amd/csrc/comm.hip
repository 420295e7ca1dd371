// RCCL (xGMI) communicator behind the C-ABI, for hosts that are not PyTorch. One process per GPU.
// The reference has no distributed code at all; the only exchange step the hot path has
// when ONE task's support frames are sharded over ranks is the sum of the per-class prototype partials
// ([C][D] sums + [C] counts, ~25 KB) produced by orbit_proto_configure — a latency-bound all-reduce.
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

using namespace orbit;

static ncclComm_t g_comm = nullptr;
static int g_world = 0, g_rank = -1;

extern "C" {

int orbit_comm_unique_id(void* out128) {
    ORBIT_REQUIRE(out128, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == ORBIT_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return set_err(ORBIT_ERR_HIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(out128, &id, sizeof(id));
    return ORBIT_OK;
}

int orbit_comm_init(int rank, int world, const void* unique_id) {
    ORBIT_REQUIRE(unique_id && world > 0 && rank >= 0 && rank < world, "comm_init: bad arguments");
    if (g_comm) return set_err(ORBIT_ERR_STATE, "comm_init: communicator already initialised");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&g_comm, world, id, rank);
    if (r != ncclSuccess) {
        g_comm = nullptr;
        return set_err(ORBIT_ERR_HIP, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    g_world = world, g_rank = rank;
    return ORBIT_OK;
}

int orbit_comm_world(void) { return g_world; }
int orbit_comm_rank(void) { return g_rank; }

int orbit_allreduce_sum(float* buf, size_t n, orbit_stream_t stream) {
    ORBIT_REQUIRE(buf && n > 0, "allreduce_sum: bad arguments");
    if (!g_comm) return set_err(ORBIT_ERR_STATE, "allreduce_sum: call orbit_comm_init first");
    ncclResult_t r = ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, g_comm, (hipStream_t)stream);
    if (r != ncclSuccess) return set_err(ORBIT_ERR_HIP, "ncclAllReduce: %s", ncclGetErrorString(r));
    return ORBIT_OK;
}

void orbit_comm_destroy(void) {
    if (g_comm) ncclCommDestroy(g_comm);
    g_comm = nullptr, g_world = 0, g_rank = -1;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// One-shot peer-to-peer all-reduce(SUM) for the SMALL exchange steps of the sharded episodic path:
// the prototype payload of a support-sharded task ([C][D] sums + [C] counts = 25.6 KB at C = 5, D = 1280) and the
// set-encoder embedding sums (65 floats). A ring / tree all-reduce of such a message is pure latency (RCCL: a kernel launch
// plus 2 (N-1) dependent xGMI hops); xGMI is a full point-to-point mesh inside a node, so every rank instead PUSHES its
// payload straight into a slot of every peer's inbox (posted remote stores, all links in parallel), raises a flag, waits for
// its own N flags and sums the N slots in RANK ORDER - one hop of latency, and the identical summation order on every rank
// makes the result bit-identical everywhere (what personalise_support_sharded promises: identical prototypes on all ranks).
//
// Memory: each rank owns an inbox  data[2][world][max_floats] + flag[world]  in device memory, exported to its peers with
// hipIpcGetMemHandle (the host side exchanges the 64-byte handles through whatever rendezvous it has: torch.distributed,
// MPI, files) and mapped by them with hipIpcOpenMemHandle. Epochs alternate the two data halves: a rank can start epoch
// e+1 while a slower peer still sums epoch e, and cannot reach e+2 before that peer has pushed e+1, i.e. finished e.
// Flags and payload cross the fabric with system-scope release / acquire; a spin that does not complete within
// ~4 s of GPU clock gives up and reports an error instead of hanging the device.
//
// LARGE payloads - the flat gradient bucket of the LITE step (16-45 MB) - take the sharded form
// (orbit_p2p_allreduce_sum_sharded): a direct reduce-scatter + all-gather over the mesh. The vector is cut into `world`
// shards; rank r pushes shard p of its vector into peer p's inbox (7 links busy at once, each carrying 1/world of the
// vector), peer p sums the world copies of ITS shard in rank order and pushes the sum to every peer, which copy it home:
// 2 (N-1)/N of the vector crosses each rank's links exactly as in a ring, but in 2 hops instead of 2 (N-1), and every
// element is summed once, by one rank, in one order (bit-identical on all ranks). The kernel runs P2P_GRID blocks; block b
// owns the same slice of every shard on every rank and synchronises only with block b of the peers (per-block flags), so no
// grid-wide barrier exists and a block's reduce overlaps the other blocks' pushes. The inbox is the same memory as the
// one-shot form's (per half: world shards in + world shards out = 2 * ceil(n / world) <= max_floats per slot).
constexpr int P2P_GRID = 64;      // blocks of the sharded form (all co-resident: no block waits on an unscheduled one)
struct orbit_p2p {
    int rank = 0, world = 0;
    size_t max_floats = 0;
    float* inbox = nullptr;                 // this rank's inbox (device)
    unsigned* flags = nullptr;              // = inbox + 2 * world * max_floats
    std::vector<float*> peer_inbox;         // mapped inbox of every rank (own entry = inbox)
    float** d_peer_inbox = nullptr;         // device copy of the table
    int* h_error = nullptr;                 // pinned, host-mapped error word (the host reads it without a device sync)
    int* d_error = nullptr;                 // its device address
    int memory_kind = 0;                    // ORBIT_P2P_MEM_*: how the inbox was allocated
    unsigned epoch = 0;
    bool connected = false;
};

namespace orbit {

// A flag holds the epoch of the sender's LATEST push. The waiter accepts any value at or past its own epoch (wrap-safe
// signed distance): a waiter that is descheduled while the sender already posts epoch e+1 must not miss e -
// the data of epoch e is still intact then, because e+1 goes to the other half and e+2 cannot start before this rank has
// pushed e+1, i.e. finished e.
__device__ __forceinline__ bool flag_reached(const unsigned* f, unsigned epoch) {
    return (int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) >= 0;
}
constexpr long long P2P_TIMEOUT_TICKS = 400000000LL;  // ~4 s at the 100 MHz constant clock

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(float* const* __restrict__ peer_inbox, float* __restrict__ buf,
                                                             int n, int rank, int world, size_t max_floats,
                                                             unsigned epoch, int* __restrict__ error) {
    const int tid = threadIdx.x;
    const size_t half = (size_t)(epoch & 1u) * world * max_floats;
    // ---- push: my payload into slot [rank] of every inbox (own included)
    for (int p = 0; p < world; ++p) {
        float* dst = peer_inbox[p] + half + (size_t)rank * max_floats;
        for (int i = tid; i < n; i += blockDim.x) __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world) {
        unsigned* f = reinterpret_cast<unsigned*>(peer_inbox[tid] + 2 * (size_t)world * max_floats) + rank;
        __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait for the world flags of my inbox
    int timed_out = 0;
    if (tid < world) {
        const unsigned* f = reinterpret_cast<const unsigned*>(peer_inbox[rank] + 2 * (size_t)world * max_floats) + tid;
        const long long t0 = wall_clock64();
        while (!flag_reached(f, epoch)) {
            if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) {
                __hip_atomic_exchange(error, 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    timed_out = __syncthreads_or(timed_out);
    __threadfence_system();
    if (timed_out) {  // a peer never arrived: the result must not look like a sum - poison it
        for (int i = tid; i < n; i += blockDim.x) buf[i] = __builtin_nanf("");
        return;
    }
    // ---- sum the slots in rank order
    const float* mine = peer_inbox[rank] + half;
    for (int i = tid; i < n; i += blockDim.x) {
        float s = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int p = 1; p < world; ++p)
            s += __hip_atomic_load(mine + (size_t)p * max_floats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = s;
    }
}

// 8-byte system-scope accesses: coherent across agents without an L2 write-back / invalidate of the whole cache
__device__ __forceinline__ void store_sys2(float* p, float a, float b) {
    unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float2 load_sys2(const float* p) {
    const unsigned long long v =
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// wait until the `world` flags f[q * stride] have all reached `epoch`; on a timeout: error code base + q, returns true
// (block-uniform)
__device__ __forceinline__ bool wait_flags(const unsigned* f, int stride, int world, unsigned epoch, int* error, int base) {
    int timed_out = 0;
    if ((int)threadIdx.x < world) {
        const unsigned* fq = f + (size_t)threadIdx.x * stride;
        const long long t0 = wall_clock64();
        while (!flag_reached(fq, epoch)) {
            if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) {
                __hip_atomic_exchange(error, base + (int)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    timed_out = __syncthreads_or(timed_out);
    __threadfence_system();
    return timed_out != 0;
}

// n floats in `buf`, shard length s (even), per-block slice length c (even): block b covers [b*c, min((b+1)*c, s)) of
// every shard. Inbox half: in[q][s] (rank q's copy of MY shard) then out[q][s] (rank q's reduced shard).
__global__ __launch_bounds__(512) void p2p_allreduce_sharded_kernel(float* const* __restrict__ peer_inbox,
                                                                    float* __restrict__ buf, size_t n, size_t s, size_t c,
                                                                    int rank, int world, size_t max_floats, unsigned epoch,
                                                                    int* __restrict__ error) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const size_t half = (size_t)(epoch & 1u) * world * max_floats;
    const size_t flag_off = 2 * (size_t)world * max_floats;  // floats; one-shot flags [world] come first
    const size_t lo = (size_t)b * c, hi = lo + c < s ? lo + c : s;
    auto flag_in = [&](int owner, int sender) {   // raised in `owner`'s inbox by `sender`: its copy of slice b arrived
        return reinterpret_cast<unsigned*>(peer_inbox[owner] + flag_off) + world + ((size_t)sender * P2P_GRID + b);
    };
    auto flag_out = [&](int owner, int sender) {  // `sender`'s reduced slice b arrived in `owner`'s inbox
        return reinterpret_cast<unsigned*>(peer_inbox[owner] + flag_off) + world + (size_t)world * P2P_GRID +
               ((size_t)sender * P2P_GRID + b);
    };
    // ---- reduce-scatter push: slice b of shard p goes to rank p (starting with my right neighbour: all links busy)
    for (int k = 0; k < world; ++k) {
        const int p = (rank + 1 + k) % world;
        float* dst = peer_inbox[p] + half + (size_t)rank * s;
        const size_t g0 = (size_t)p * s;
        for (size_t i = lo + 2 * (size_t)tid; i < hi; i += 2 * blockDim.x) {
            const size_t g = g0 + i;
            store_sys2(dst + i, g < n ? buf[g] : 0.f, g + 1 < n ? buf[g + 1] : 0.f);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world) __hip_atomic_store(flag_in(tid, rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- my shard: sum the world copies of slice b in rank order, push the sum to every rank (own inbox included)
    auto poison = [&]() {  // slice b of every shard of buf: a timed-out exchange must not look like a sum
        for (int q = 0; q < world; ++q)
            for (size_t i = lo + tid; i < hi; i += blockDim.x)
                if ((size_t)q * s + i < n) buf[(size_t)q * s + i] = __builtin_nanf("");
    };
    if (wait_flags(flag_in(rank, 0), P2P_GRID, world, epoch, error, 1)) {
        poison();
        return;
    }
    const float* in = peer_inbox[rank] + half;
    for (size_t i = lo + 2 * (size_t)tid; i < hi; i += 2 * blockDim.x) {
        float2 acc = load_sys2(in + i);
        for (int q = 1; q < world; ++q) {
            const float2 v = load_sys2(in + (size_t)q * s + i);
            acc.x += v.x, acc.y += v.y;
        }
        for (int k = 0; k < world; ++k) {
            const int p = (rank + 1 + k) % world;
            store_sys2(peer_inbox[p] + half + (size_t)(world + rank) * s + i, acc.x, acc.y);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world) __hip_atomic_store(flag_out(tid, rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- all-gather: copy slice b of every reduced shard home
    if (wait_flags(flag_out(rank, 0), P2P_GRID, world, epoch, error, 101)) {
        poison();
        return;
    }
    const float* out = peer_inbox[rank] + half + (size_t)world * s;
    for (int q = 0; q < world; ++q) {
        const size_t g0 = (size_t)q * s;
        for (size_t i = lo + 2 * (size_t)tid; i < hi; i += 2 * blockDim.x) {
            const float2 v = load_sys2(out + (size_t)q * s + i);
            const size_t g = g0 + i;
            if (g < n) buf[g] = v.x;
            if (g + 1 < n) buf[g + 1] = v.y;
        }
    }
}

}  // namespace orbit

extern "C" {

int orbit_p2p_create(int rank, int world, size_t max_floats, orbit_p2p_t** out) {
    ORBIT_REQUIRE(out && world > 0 && world <= 64 && rank >= 0 && rank < world && max_floats > 0 && max_floats <= (1u << 26),
                  "p2p_create: bad arguments (rank %d, world %d, max_floats %zu)", rank, world, max_floats);
    orbit_p2p* c = new orbit_p2p();
    c->rank = rank, c->world = world, c->max_floats = (max_floats + 3) & ~(size_t)3;
    const size_t bytes = 2 * (size_t)world * c->max_floats * sizeof(float) +
                         (size_t)world * (1 + 2 * P2P_GRID) * sizeof(unsigned);  // one-shot flags, then the sharded form's
    // The inbox is written by peers and polled by this rank WHILE a kernel runs: it must not be ordinary (coarse-grained)
    // device memory, which is only coherent across agents at kernel boundaries - over xGMI the owner's L2 may keep serving
    // stale flag / payload lines. Uncached device memory (MTYPE_UC, what RCCL uses for its signal buffers on
    // gfx942 / gfx950), else fine-grained; coarse-grained only when ORBIT_P2P_ALLOW_COARSE=1 (single-GPU experiments).
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->inbox), bytes, hipDeviceMallocUncached);
    c->memory_kind = ORBIT_P2P_MEM_UNCACHED;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->inbox), bytes, hipDeviceMallocFinegrained);
        c->memory_kind = ORBIT_P2P_MEM_FINEGRAINED;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        const char* allow = getenv("ORBIT_P2P_ALLOW_COARSE");
        if (allow && allow[0] == '1') {
            e = hipMalloc(reinterpret_cast<void**>(&c->inbox), bytes);
            c->memory_kind = ORBIT_P2P_MEM_COARSE;
        }
    }
    if (e == hipSuccess) e = hipMemset(c->inbox, 0, bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_peer_inbox), world * sizeof(float*));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_error), sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->h_error = 0;
        e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_error), c->h_error, 0);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(c->inbox), (void)hipFree(c->d_peer_inbox);
        if (c->h_error) (void)hipHostFree(c->h_error);
        delete c;
        (void)hipGetLastError();
        return set_err(ORBIT_ERR_HIP, "p2p_create: %s (inbox of %zu bytes, uncached / fine-grained device memory)",
                       hipGetErrorString(e), bytes);
    }
    c->flags = reinterpret_cast<unsigned*>(c->inbox + 2 * (size_t)world * c->max_floats);
    c->peer_inbox.assign(world, nullptr);
    c->peer_inbox[rank] = c->inbox;
    *out = c;
    return ORBIT_OK;
}

int orbit_p2p_export(orbit_p2p_t* c, void* handle64) {
    ORBIT_REQUIRE(c && handle64, "p2p_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == ORBIT_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t h;
    ORBIT_HIP_CHECK(hipIpcGetMemHandle(&h, c->inbox));
    memcpy(handle64, &h, sizeof(h));
    return ORBIT_OK;
}

int orbit_p2p_connect(orbit_p2p_t* c, const void* handles) {
    ORBIT_REQUIRE(c && handles, "p2p_connect: null pointer");
    if (c->connected) return set_err(ORBIT_ERR_STATE, "p2p_connect: already connected");
    const char* hs = static_cast<const char*>(handles);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, hs + (size_t)p * ORBIT_P2P_HANDLE_BYTES, sizeof(h));
        void* ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return set_err(ORBIT_ERR_HIP, "p2p_connect: hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e));
        }
        c->peer_inbox[p] = static_cast<float*>(ptr);
    }
    ORBIT_HIP_CHECK(hipMemcpy(c->d_peer_inbox, c->peer_inbox.data(), c->world * sizeof(float*), hipMemcpyHostToDevice));
    c->connected = true;
    return ORBIT_OK;
}

int orbit_p2p_allreduce_sum(orbit_p2p_t* c, float* buf, size_t n, orbit_stream_t stream) {
    ORBIT_REQUIRE(c && buf && n > 0, "p2p_allreduce_sum: bad arguments");
    ORBIT_REQUIRE(c->connected, "p2p_allreduce_sum: call orbit_p2p_connect first");
    ORBIT_REQUIRE(n <= c->max_floats, "p2p_allreduce_sum: %zu floats exceed the inbox slot (%zu)", n, c->max_floats);
    ++c->epoch;  // flags start at 0 = "epoch 0 reached"; the wait is a wrap-safe signed distance, so any sequence works
    const int threads = n >= 4096 ? 1024 : 256;
    orbit::p2p_allreduce_kernel<<<1, threads, 0, (hipStream_t)stream>>>(c->d_peer_inbox, buf, (int)n, c->rank, c->world,
                                                                       c->max_floats, c->epoch, c->d_error);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

/* Large payloads: direct reduce-scatter + all-gather over the mesh (see the header of this section). n <= world * max_floats / 2. */
int orbit_p2p_allreduce_sum_sharded(orbit_p2p_t* c, float* buf, size_t n, orbit_stream_t stream) {
    ORBIT_REQUIRE(c && buf && n > 0, "p2p_allreduce_sum_sharded: bad arguments");
    ORBIT_REQUIRE(c->connected, "p2p_allreduce_sum_sharded: call orbit_p2p_connect first");
    ORBIT_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 3) == 0, "p2p_allreduce_sum_sharded: unaligned buffer");
    size_t s = (n + c->world - 1) / c->world;
    s = (s + 1) & ~(size_t)1;  // 8-byte accesses
    ORBIT_REQUIRE(2 * s <= c->max_floats, "p2p_allreduce_sum_sharded: %zu floats need inbox slots of %zu floats (have %zu)", n,
                  2 * s, c->max_floats);
    size_t slice = (s + P2P_GRID - 1) / P2P_GRID;
    slice = (slice + 1) & ~(size_t)1;
    ++c->epoch;
    orbit::p2p_allreduce_sharded_kernel<<<P2P_GRID, 512, 0, (hipStream_t)stream>>>(
        c->d_peer_inbox, buf, n, s, slice, c->rank, c->world, c->max_floats, c->epoch, c->d_error);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

/* 0 = no flag wait has timed out so far; k > 0 = a wait for rank (k-1) % 100's flag timed out and that all-reduce's buffer
 * was filled with NaN. Reads a host-mapped word: does NOT synchronise the device, so it reports the all-reduces that have
 * COMPLETED by now (call it after a sync point, or once per optimizer step about the step before). */
int orbit_p2p_error(orbit_p2p_t* c) {
    ORBIT_REQUIRE(c, "p2p_error: null pointer");
    return __atomic_load_n(c->h_error, __ATOMIC_ACQUIRE);
}

int orbit_p2p_memory_kind(orbit_p2p_t* c) { return c ? c->memory_kind : 0; }

void orbit_p2p_destroy(orbit_p2p_t* c) {
    if (!c) return;
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank && c->peer_inbox[p]) (void)hipIpcCloseMemHandle(c->peer_inbox[p]);
    (void)hipFree(c->inbox), (void)hipFree(c->d_peer_inbox);
    if (c->h_error) (void)hipHostFree(c->h_error);
    delete c;
}

}  // extern "C"
