// cvhip_rccl.hip — the row-sharding collectives of libcvhip.so on RCCL, without torch.
//
// north_star: "the disparity map shards across the 8 GPUs of one node with a single RCCL gather over xGMI".
// One process per GPU; each process creates a communicator from a 128-byte id that rank 0 makes and the launcher
// distributes (torch.distributed's store in bench.py, MPI or a file for the Rust host).  Every collective is
// enqueued on the DEVICE HANDLE'S OWN STREAM, so it is ordered with the search kernels before it and the
// cross-check kernels after it with no host synchronisation and no reliance on the caller's stream discipline.
// Only concatenation crosses ranks (no reduction): N-GPU output is bit-identical to 1-GPU output.
// librccl.so.1 is opened lazily with dlopen, so a single-GPU user never needs it; when torch is in the process its
// bundled librccl (same soname) is the one that gets used, and a process never holds two RCCL runtimes.
#include "cvhip_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

struct cvhip_rccl {
    cvhip_device *dev = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
};

namespace {

struct Api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Api &api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) {
            a.error = std::string("cannot load librccl.so.1: ") + dlerror();
            return;
        }
        const auto sym = [&](const char *n) {
            void *p = dlsym(a.handle, n);
            if (!p && a.error.empty()) a.error = std::string("librccl has no symbol ") + n;
            return p;
        };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return a;
}

int need_api()
{
    Api &a = api();
    if (!a.error.empty()) return cvhip::fail(CVHIP_ERR_UNSUPPORTED, a.error);
    return CVHIP_OK;
}

int nccl_fail(const char *what, ncclResult_t r)
{
    const char *msg = api().GetErrorString ? api().GetErrorString(r) : "?";
    return cvhip::fail(CVHIP_ERR_DEVICE, std::string(what) + ": " + msg);
}

#define CVHIP_TRY_NCCL(expr)                              \
    do {                                                  \
        const ncclResult_t _r = (expr);                   \
        if (_r != ncclSuccess) return nccl_fail(#expr, _r); \
    } while (0)

// the all-gather hook cvhip_correlate_level calls after a sharded search pass (cvhip_allgather_fn)
int rccl_hook(void *user, void *cells, uint64_t shard_bytes, uint32_t n_shards, int /*dir*/)
{
    cvhip_rccl *c = static_cast<cvhip_rccl *>(user);
    if (!c || n_shards != c->world) return 1;
    return cvhip_rccl_allgather(c, cells, shard_bytes) == CVHIP_OK ? 0 : 1;
}

} // namespace

using namespace cvhip;

extern "C" {

int cvhip_rccl_unique_id(uint8_t *id)
{
    if (!id) return fail(CVHIP_ERR_INVALID, "id is null");
    CVHIP_TRY(need_api());
    static_assert(sizeof(ncclUniqueId) == CVHIP_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    CVHIP_TRY_NCCL(api().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return CVHIP_OK;
}

int cvhip_rccl_create(cvhip_device *dev, const uint8_t *id, uint32_t rank, uint32_t world, cvhip_rccl **out)
{
    if (!dev || !id || !out) return fail(CVHIP_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world == 0 || world > 64 || rank >= world) return fail(CVHIP_ERR_INVALID, "need 0 <= rank < world <= 64");
    CVHIP_TRY(need_api());
    CVHIP_TRY_HIP(hipSetDevice(dev->d.ordinal));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    cvhip_rccl *c = new (std::nothrow) cvhip_rccl();
    if (!c) return fail(CVHIP_ERR_NOMEM, "out of host memory");
    c->dev = dev;
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = api().CommInitRank(&c->comm, (int)world, u, (int)rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail("ncclCommInitRank", r);
    }
    dev->d.comm_refs++; // the communicator uses the handle's stream: the handle outlives it (cvhip_device_destroy defers)
    *out = c;
    return CVHIP_OK;
}

void cvhip_rccl_destroy(cvhip_rccl *comm)
{
    if (!comm) return;
    (void)hipSetDevice(comm->dev->d.ordinal);
    (void)hipStreamSynchronize(comm->dev->d.stream);
    if (comm->comm && api().CommDestroy) (void)api().CommDestroy(comm->comm);
    cvhip_device *dev = comm->dev;
    delete comm;
    // a cvhip_device_destroy that arrived while this communicator was alive was deferred: finish it now
    if (--dev->d.comm_refs == 0 && dev->d.destroy_pending) cvhip::device_free(dev);
}

int cvhip_rccl_allgather(cvhip_rccl *comm, void *buf, uint64_t shard_bytes)
{
    if (!comm || !buf) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY_HIP(hipSetDevice(comm->dev->d.ordinal));
    // in place: rank r's chunk already sits at buf + r * shard_bytes (NCCL's in-place convention)
    uint8_t *base = static_cast<uint8_t *>(buf);
    CVHIP_TRY_NCCL(api().AllGather(base + (size_t)comm->rank * shard_bytes, base, (size_t)shard_bytes, ncclUint8, comm->comm,
                                   comm->dev->d.stream));
    return CVHIP_OK;
}

int cvhip_rccl_gather(cvhip_rccl *comm, void *buf, uint64_t shard_bytes, uint32_t root)
{
    if (!comm || !buf) return fail(CVHIP_ERR_INVALID, "null argument");
    if (root >= comm->world) return fail(CVHIP_ERR_INVALID, "root out of range");
    CVHIP_TRY_HIP(hipSetDevice(comm->dev->d.ordinal));
    uint8_t *base = static_cast<uint8_t *>(buf);
    hipStream_t s = comm->dev->d.stream;
    // one grouped exchange: every other rank sends its chunk straight into the root's buffer over its own xGMI
    // link (7 concurrent point-to-point transfers on an 8-GPU node; no ring).  A world of one exchanges its chunk with
    // itself in place - the same grouped ncclSend / ncclRecv, so a single-GPU box executes this path too.
    const bool self = comm->world == 1;
    CVHIP_TRY_NCCL(api().GroupStart());
    ncclResult_t r = ncclSuccess;
    if (comm->rank == root) {
        for (uint32_t p = 0; p < comm->world && r == ncclSuccess; p++)
            if (p != root || self) r = api().Recv(base + (size_t)p * shard_bytes, (size_t)shard_bytes, ncclUint8, (int)p, comm->comm, s);
    }
    if ((comm->rank != root || self) && r == ncclSuccess)
        r = api().Send(base + (size_t)comm->rank * shard_bytes, (size_t)shard_bytes, ncclUint8, (int)root, comm->comm, s);
    const ncclResult_t e = api().GroupEnd();
    if (r != ncclSuccess) return nccl_fail("ncclSend/ncclRecv", r);
    if (e != ncclSuccess) return nccl_fail("ncclGroupEnd", e);
    return CVHIP_OK;
}

int cvhip_ctx_set_row_shard_rccl(cvhip_ctx *ctx, cvhip_rccl *comm)
{
    if (!ctx || !comm) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(cvhip::flush_level_calls(ctx));
    if (ctx->dev != comm->dev) return fail(CVHIP_ERR_INVALID, "context and communicator belong to different device handles");
    CVHIP_TRY(cvhip_ctx_set_row_shard(ctx, comm->rank, comm->world, rccl_hook, comm));
    ctx->gather_on_stream = true;
    return CVHIP_OK;
}

int cvhip_ctx_gather_bands_rccl(cvhip_ctx *ctx, cvhip_rccl *comm, int root)
{
    if (!ctx || !comm) return fail(CVHIP_ERR_INVALID, "null argument");
    CVHIP_TRY(cvhip::flush_level_calls(ctx));
    if (ctx->dev != comm->dev) return fail(CVHIP_ERR_INVALID, "context and communicator belong to different device handles");
    if (ctx->shard_den != comm->world || ctx->shard_num != comm->rank)
        return fail(CVHIP_ERR_INVALID, "the context's shard is not the communicator's rank");
    void *cells = nullptr, *scores = nullptr;
    uint32_t lw = 0, lh = 0, rps = 0;
    CVHIP_TRY(cvhip_ctx_level_grid(ctx, 0, &cells, &scores, &lw, &lh, nullptr, nullptr, &rps));
    // the two planes of the forward grid: matches (u32) and scores (f32), 4 + 4 bytes per level pixel
    const uint64_t shard_bytes = (uint64_t)rps * lw * sizeof(uint32_t);
    if (root >= (int)comm->world) return fail(CVHIP_ERR_INVALID, "root out of range");
    if (root < 0) {
        CVHIP_TRY(cvhip_rccl_allgather(comm, cells, shard_bytes));
        return cvhip_rccl_allgather(comm, scores, shard_bytes);
    }
    CVHIP_TRY(cvhip_rccl_gather(comm, cells, shard_bytes, (uint32_t)root));
    return cvhip_rccl_gather(comm, scores, shard_bytes, (uint32_t)root);
}

} // extern "C"
