// RCCL (xGMI) communicator behind the C-ABI, for hosts that are not PyTorch. One process per GPU.
// The reference has no distributed code at all; the only exchange step the hot path has
// when ONE task's support frames are sharded over ranks is the sum of the per-class prototype partials
// ([C][D] sums + [C] counts, ~25 KB) produced by orbit_proto_configure — a latency-bound all-reduce.
#include <rccl/rccl.h>
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
