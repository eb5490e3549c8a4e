// capi_comm.hip -- C ABI (include/gmmiv.h): the collectives of the paths that shard, on RCCL over xGMI.
//
// One communicator per context (= per GPU, per process or per host thread).  The reference merges the private accumulators
// of its worker threads under a mutex (LIA_SpkTools/src/AccumulateStat.cpp:286-292 addAccEM; AccumulateTVStat.cpp:1920-1937,
// 2036-2044 `+=` of A / Cmx / R / r); here the ranks' accumulators are summed by ONE collective on the device buffers.
// RCCL is resolved at run time (dlopen) so that libgmmiv.so keeps a single hard dependency, the HIP runtime: a single-GPU
// user never needs RCCL (world == 1 is handled without it), and inside a PyTorch process the copy of RCCL that torch already
// mapped is reused instead of loading a second one next to it.
#include <dlfcn.h>
#include <link.h>
#include <stdlib.h>
#include <unistd.h>

#include <mutex>

#include <rccl/rccl.h>

#include "ctx.h"

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    std::string where;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_err;

int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        *(std::string *)data = info->dlpi_name;
        return 1;
    }
    return 0;
}

void load_rccl()
{
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded); // e.g. torch/lib/librccl.so inside a PyTorch process
    const char *env = getenv("GMMIV_RCCL_LIB");
    const char *cand[] = {env, loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *name : cand) {
        if (!name || !*name) continue;
        g_rccl.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) { g_rccl.where = name; break; }
    }
    if (!g_rccl.h) { g_rccl_err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?"); return; }
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(g_rccl.h, n); if (!p) { ok = false; g_rccl_err = std::string("RCCL symbol missing: ") + n; } return p; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))sym("ncclReduceScatter");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    if (!ok) { dlclose(g_rccl.h); g_rccl.h = nullptr; }
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.h) { gmmiv_set_error("%s", g_rccl_err.c_str()); return nullptr; }
    return &g_rccl;
}

} // namespace

struct gmmiv_comm {
    gmmiv_ctx *ctx = nullptr;
    int world = 1, rank = 0;
    ncclComm_t nc = nullptr;
    const Rccl *api = nullptr;
    void *stage = nullptr; // device staging for HOST buffers (grow-only)
    size_t stage_bytes = 0;
    double bytes_moved = 0.0; // payload bytes this rank handed to collectives since the last query
    int staged(size_t bytes, void **out)
    {
        if (stage_bytes < bytes) {
            if (stage) { GCHK(hipStreamSynchronize(ctx->stream)); GCHK(hipFree(stage)); stage = nullptr; stage_bytes = 0; }
            GCHK(hipMalloc(&stage, bytes));
            stage_bytes = bytes;
        }
        *out = stage;
        return GMMIV_OK;
    }
};

// Release everything that needs the context (called by gmmiv_comm_destroy, and by gmmiv_ctx_destroy for communicators the
// caller still holds: afterwards the handle is an empty shell that gmmiv_comm_destroy can still delete safely).
void gmmiv_comm_orphan(gmmiv_comm *c)
{
    if (!c || !c->ctx) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->nc) { (void)c->api->CommDestroy(c->nc); c->nc = nullptr; }
    if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
    auto &v = c->ctx->comms;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == c) { v.erase(v.begin() + i); break; }
    c->ctx = nullptr;
}

#define NCHK(c, expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t _r = (expr);                                                                                      \
        if (_r != ncclSuccess) {                                                                                       \
            gmmiv_set_error("%s:%d: %s -> %s (rank %d of %d)", __FILE__, __LINE__, #expr, (c)->api->GetErrorString(_r), (c)->rank, (c)->world); \
            return GMMIV_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)

extern "C" {

void gmmiv_shard_range(int64_t n, int rank, int world, int64_t *begin, int64_t *end)
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    const int64_t base = n / world, rem = n % world;
    const int64_t b = rank * base + (rank < rem ? rank : rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

int gmmiv_comm_get_unique_id(void *id128)
{
    if (!id128) { gmmiv_set_error("comm_get_unique_id: id == NULL"); return GMMIV_ERR_ARG; }
    const Rccl *api = rccl();
    if (!api) return GMMIV_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == GMMIV_COMM_ID_BYTES, "gmmiv.h and rccl.h disagree on the id size");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { gmmiv_set_error("ncclGetUniqueId -> %s", api->GetErrorString(r)); return GMMIV_ERR_HIP; }
    memcpy(id128, &id, sizeof(id));
    return GMMIV_OK;
}

int gmmiv_comm_exchange_id_file(const char *path, int rank, void *id128, double timeout_s)
{
    if (!path || !id128 || rank < 0) { gmmiv_set_error("comm_exchange_id_file: bad argument"); return GMMIV_ERR_ARG; }
    if (rank == 0) {
        int rc = gmmiv_comm_get_unique_id(id128);
        if (rc) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id128, 1, GMMIV_COMM_ID_BYTES, f) != GMMIV_COMM_ID_BYTES) { if (f) fclose(f); gmmiv_set_error("comm_exchange_id_file: cannot write %s", tmp.c_str()); return GMMIV_ERR_ARG; }
        fclose(f);
        if (rename(tmp.c_str(), path) != 0) { gmmiv_set_error("comm_exchange_id_file: cannot rename %s", tmp.c_str()); return GMMIV_ERR_ARG; }
        return GMMIV_OK;
    }
    const double step = 0.01;
    for (double waited = 0.0; waited <= timeout_s; waited += step) { // the rename above makes the file appear complete
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t n = fread(id128, 1, GMMIV_COMM_ID_BYTES, f);
            fclose(f);
            if (n == GMMIV_COMM_ID_BYTES) return GMMIV_OK;
        }
        usleep((useconds_t)(step * 1e6));
    }
    gmmiv_set_error("comm_exchange_id_file: rank %d waited %.0f s for %s", rank, timeout_s, path);
    return GMMIV_ERR_HIP;
}

int gmmiv_comm_create(gmmiv_ctx *ctx, int world, int rank, const void *id128, gmmiv_comm **out)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) { gmmiv_set_error("comm_create: bad argument"); return GMMIV_ERR_ARG; }
    GCHK(hipSetDevice(ctx->device));
    gmmiv_comm *c = new gmmiv_comm();
    c->ctx = ctx; c->world = world; c->rank = rank;
    if (world > 1) {
        c->api = rccl();
        if (!c->api) { delete c; return GMMIV_ERR_UNSUPPORTED; }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        ncclResult_t r = c->api->CommInitRank(&c->nc, world, id, rank);
        if (r != ncclSuccess) {
            gmmiv_set_error("ncclCommInitRank(world %d, rank %d, device %d) -> %s", world, rank, ctx->device, c->api->GetErrorString(r));
            delete c;
            return GMMIV_ERR_HIP;
        }
    }
    ctx->comms.push_back(c);
    *out = c;
    return GMMIV_OK;
}

void gmmiv_comm_destroy(gmmiv_comm *c)
{
    if (!c) return;
    gmmiv_comm_orphan(c); // releases RCCL and the staging buffer (no-op when the context already went away)
    delete c;
}

int gmmiv_comm_world(const gmmiv_comm *c) { return c ? c->world : 0; }
int gmmiv_comm_rank(const gmmiv_comm *c) { return c ? c->rank : -1; }
const char *gmmiv_comm_backend(const gmmiv_comm *c)
{
    if (!c) return "";
    return c->world == 1 || !c->api ? "single rank (no collective library)" : c->api->where.c_str();
}
double gmmiv_comm_take_bytes(gmmiv_comm *c)
{
    if (!c) return 0.0;
    const double b = c->bytes_moved;
    c->bytes_moved = 0.0;
    return b;
}

// buf[n] <- sum over ranks, in place; buf: host or device
int gmmiv_allreduce_f64(gmmiv_comm *c, double *buf, size_t n)
{
    if (!c || !c->ctx || (!buf && n)) { gmmiv_set_error("allreduce_f64: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)n * 8;
    if (c->world == 1 || n == 0) return GMMIV_OK;
    GCHK(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    if (gmmiv_is_device_ptr(buf)) {
        NCHK(c, c->api->AllReduce(buf, buf, n, ncclFloat64, ncclSum, c->nc, st));
        return GMMIV_OK;
    }
    void *d;
    int rc = c->staged(n * 8, &d);
    if (rc) return rc;
    GCHK(hipMemcpyAsync(d, buf, n * 8, hipMemcpyHostToDevice, st));
    NCHK(c, c->api->AllReduce(d, d, n, ncclFloat64, ncclSum, c->nc, st));
    GCHK(hipMemcpyAsync(buf, d, n * 8, hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    return GMMIV_OK;
}

// recv[recvcount] <- block `rank` of the sum over ranks of send[world * recvcount]; DEVICE buffers.
// In place when recv == send + rank * recvcount.
int gmmiv_reduce_scatter_f64(gmmiv_comm *c, const double *send, double *recv, size_t recvcount)
{
    if (!c || !c->ctx || ((!send || !recv) && recvcount)) { gmmiv_set_error("reduce_scatter_f64: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)recvcount * 8 * c->world;
    if (recvcount == 0) return GMMIV_OK;
    if (!gmmiv_is_device_ptr(send) || !gmmiv_is_device_ptr(recv)) { gmmiv_set_error("reduce_scatter_f64: device buffers only"); return GMMIV_ERR_ARG; }
    GCHK(hipSetDevice(c->ctx->device));
    if (c->world == 1) {
        if (recv != send) GCHK(hipMemcpyAsync(recv, send, recvcount * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
        return GMMIV_OK;
    }
    NCHK(c, c->api->ReduceScatter(send, recv, recvcount, ncclFloat64, ncclSum, c->nc, c->ctx->stream));
    return GMMIV_OK;
}

// recv[world * sendcount] <- the ranks' send[sendcount] in rank order; DEVICE buffers.  In place when send == recv + rank * sendcount.
int gmmiv_allgather_f64(gmmiv_comm *c, const double *send, double *recv, size_t sendcount)
{
    if (!c || !c->ctx || ((!send || !recv) && sendcount)) { gmmiv_set_error("allgather_f64: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)sendcount * 8 * c->world;
    if (sendcount == 0) return GMMIV_OK;
    if (!gmmiv_is_device_ptr(send) || !gmmiv_is_device_ptr(recv)) { gmmiv_set_error("allgather_f64: device buffers only"); return GMMIV_ERR_ARG; }
    GCHK(hipSetDevice(c->ctx->device));
    if (c->world == 1) {
        if (recv != send) GCHK(hipMemcpyAsync(recv, send, sendcount * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
        return GMMIV_OK;
    }
    NCHK(c, c->api->AllGather(send, recv, sendcount, ncclFloat64, c->nc, c->ctx->stream));
    return GMMIV_OK;
}

// buf[n] on every rank <- buf of `root`; host or device
int gmmiv_broadcast_f64(gmmiv_comm *c, double *buf, size_t n, int root)
{
    if (!c || !c->ctx || (!buf && n) || root < 0 || root >= c->world) { gmmiv_set_error("broadcast_f64: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)n * 8;
    if (c->world == 1 || n == 0) return GMMIV_OK;
    GCHK(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    if (gmmiv_is_device_ptr(buf)) {
        NCHK(c, c->api->Broadcast(buf, buf, n, ncclFloat64, root, c->nc, st));
        return GMMIV_OK;
    }
    void *d;
    int rc = c->staged(n * 8, &d);
    if (rc) return rc;
    if (c->rank == root) GCHK(hipMemcpyAsync(d, buf, n * 8, hipMemcpyHostToDevice, st));
    NCHK(c, c->api->Broadcast(d, d, n, ncclFloat64, root, c->nc, st));
    GCHK(hipMemcpyAsync(buf, d, n * 8, hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    return GMMIV_OK;
}

} // extern "C"
