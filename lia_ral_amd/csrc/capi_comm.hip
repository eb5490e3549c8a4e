// capi_comm.hip -- C ABI (include/gmmiv.h): the collectives of the paths that shard, on RCCL over xGMI.
//
// One communicator per context (= per GPU, per process or per host thread).  The reference merges the private accumulators
// of its worker threads under a mutex (LIA_SpkTools/src/AccumulateStat.cpp:286-292 addAccEM; AccumulateTVStat.cpp:1920-1937,
// 2036-2044 `+=` of A / Cmx / R / r); here the ranks' accumulators are summed by ONE collective on the device buffers.
// RCCL is resolved at run time (dlopen) so that libgmmiv.so keeps a single hard dependency, the HIP runtime: a single-GPU
// user never needs RCCL (world == 1 is handled without it), and inside a PyTorch process the copy of RCCL that torch already
// mapped is reused instead of loading a second one next to it.
//
// Second transport, "shm" (opt-in, chosen by the id rank 0 draws): the ranks stage their buffers through ONE mmap'ed file and
// every rank sums the pieces ON ITS DEVICE in rank order (bitwise the same result on every rank).  It exists for ranks that
// SHARE a GPU (RCCL refuses two ranks on one device) and for boxes without RCCL: it is how the multi-rank orchestration of
// the T-matrix EM / UBM EM runs on a one-GPU machine.  A production node uses RCCL over xGMI.
#include <dlfcn.h>
#include <fcntl.h>
#include <link.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
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
    ncclResult_t (*GetVersion)(int *) = nullptr;            // optional: reported by gmmiv_comm_info
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
    // GMMIV_RCCL_LIB, when set, is the ONLY candidate (an override that silently fell through to another copy would hide a typo)
    const char *cand_env[] = {env};
    const char *cand_def[] = {loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const bool forced = env && *env;
    const char *const *cand = forced ? cand_env : cand_def;
    const int ncand = forced ? 1 : 4;
    for (int ci = 0; ci < ncand; ++ci) {
        const char *name = cand[ci];
        if (!name || !*name) continue;
        g_rccl.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) { g_rccl.where = name; break; }
    }
    if (!g_rccl.h) {
        const char *e = dlerror(); // ONE call: dlerror() clears the message it returns
        g_rccl_err = std::string("RCCL not found (dlopen librccl.so.1): ") + (e ? e : "?");
        return;
    }
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
    if (!ok) { dlclose(g_rccl.h); g_rccl.h = nullptr; return; }
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(g_rccl.h, "ncclGetVersion");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.h, "ncclCommCount");
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.h) { gmmiv_set_error("%s", g_rccl_err.c_str()); return nullptr; }
    return &g_rccl;
}

} // namespace

// ---- "shm" transport: one mmap'ed file, a slot per rank, a sense-reversing barrier in its header ----------------------------
static const char SHM_MAGIC[8] = {'G', 'M', 'M', 'I', 'V', 'S', 'H', 'M'};

struct ShmHdr {
    char magic[8];
    uint32_t world;
    std::atomic<uint32_t> ready;   // rank 0 sets it once the header is initialised
    std::atomic<uint32_t> count;   // barrier arrivals of the current generation
    std::atomic<uint32_t> gen;     // barrier generation
    std::atomic<uint32_t> failed;  // a rank gave up (timeout / HIP error): the others stop waiting
    uint64_t slot_bytes;
};
static const size_t SHM_HDR_BYTES = 4096;
static_assert(sizeof(ShmHdr) <= SHM_HDR_BYTES, "header page");

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

__global__ void k_comm_add_f64(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

struct gmmiv_comm {
    gmmiv_ctx *ctx = nullptr;
    int world = 1, rank = 0;
    ncclComm_t nc = nullptr;
    const Rccl *api = nullptr;
    void *stage = nullptr; // device staging for HOST buffers (grow-only)
    size_t stage_bytes = 0;
    double bytes_moved = 0.0; // payload bytes this rank handed to collectives since the last query
    // shm transport
    ShmHdr *shm = nullptr;
    size_t shm_bytes = 0;
    std::string shm_path, backend;
    void *addbuf = nullptr; // device staging of one slot (the summand of a peer)
    double timeout_s = 300.0;
    // overlapped collectives (gmmiv_*_begin / gmmiv_comm_join): a side stream forked from / joined to the context's stream by events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool pending = false;
    int fork_side()
    {
        if (!side) {
            GCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            GCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            GCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        GCHK(hipEventRecord(ev_fork, ctx->stream));
        GCHK(hipStreamWaitEvent(side, ev_fork, 0));
        return GMMIV_OK;
    }
    int mark_side()
    {
        GCHK(hipEventRecord(ev_join, side));
        pending = true;
        return GMMIV_OK;
    }
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
    bool is_shm() const { return shm != nullptr; }
    bool local() const { return world == 1 && !nc; } // one rank and no collective library behind it: every collective is an identity / a copy
    double *slot(int r) const { return (double *)((char *)shm + SHM_HDR_BYTES + (size_t)r * shm->slot_bytes); }
    size_t slot_elems() const { return (size_t)(shm->slot_bytes / 8); }
    int fail(const char *what)
    {
        if (shm) shm->failed.store(1, std::memory_order_release);
        gmmiv_set_error("gmmiv_comm (shm, rank %d of %d): %s", rank, world, what);
        return GMMIV_ERR_HIP;
    }
    int barrier()
    {
        const uint32_t g = shm->gen.load(std::memory_order_acquire);
        if (shm->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            shm->count.store(0, std::memory_order_relaxed);
            shm->gen.store(g + 1, std::memory_order_release);
            return GMMIV_OK;
        }
        const double t0 = now_s();
        for (unsigned spin = 0; shm->gen.load(std::memory_order_acquire) == g; ++spin) {
            if (shm->failed.load(std::memory_order_acquire)) return fail("a peer rank failed");
            if (spin > 2000) usleep(50);
            if ((spin & 1023) == 1023 && now_s() - t0 > timeout_s) return fail("barrier timed out (a peer rank is gone?)");
        }
        return GMMIV_OK;
    }
    // dst[m] (device) <- sum over ranks r = 0 .. world-1, in that order, of the m doubles at element offset `eoff` of slot r
    int sum_slots(double *dst, size_t eoff, size_t m)
    {
        hipStream_t st = ctx->stream;
        GCHK(hipMemcpyAsync(dst, slot(0) + eoff, m * 8, hipMemcpyHostToDevice, st));
        for (int r = 1; r < world; ++r) {
            GCHK(hipMemcpyAsync(addbuf, slot(r) + eoff, m * 8, hipMemcpyHostToDevice, st));
            const unsigned blocks = (unsigned)((m + 255) / 256 < 2048 ? (m + 255) / 256 : 2048);
            hipLaunchKernelGGL(k_comm_add_f64, dim3(blocks), dim3(256), 0, st, dst, (const double *)addbuf, m);
            GCHK(hipGetLastError());
        }
        GCHK(hipStreamSynchronize(st));
        return GMMIV_OK;
    }
};

#define SCHK(c, expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (hipError_t)(expr);                                                                            \
        if (_e != hipSuccess) {                                                                                        \
            (c)->shm->failed.store(1, std::memory_order_release);                                                      \
            gmmiv_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));                      \
            return GMMIV_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)
#define SRC(c, expr)                                                                                                   \
    do {                                                                                                               \
        int _rc = (expr);                                                                                              \
        if (_rc) { (c)->shm->failed.store(1, std::memory_order_release); return _rc; }                                 \
    } while (0)

namespace {

int shm_allreduce(gmmiv_comm *c, double *buf, size_t n)
{
    hipStream_t st = c->ctx->stream;
    const size_t ch = c->slot_elems();
    for (size_t off = 0; off < n; off += ch) {
        const size_t m = n - off < ch ? n - off : ch;
        SCHK(c, hipMemcpyAsync(c->slot(c->rank), buf + off, m * 8, hipMemcpyDeviceToHost, st));
        SCHK(c, hipStreamSynchronize(st));
        SRC(c, c->barrier());
        SRC(c, c->sum_slots(buf + off, 0, m));
        SRC(c, c->barrier()); // nobody overwrites its slot before everybody has read it
    }
    return GMMIV_OK;
}

int shm_reduce_scatter(gmmiv_comm *c, const double *send, double *recv, size_t recvcount)
{
    hipStream_t st = c->ctx->stream;
    const size_t ch = c->slot_elems() / (size_t)c->world; // a slot holds one piece per destination
    if (ch == 0) return c->fail("slot smaller than the world size");
    for (size_t off = 0; off < recvcount; off += ch) {
        const size_t m = recvcount - off < ch ? recvcount - off : ch;
        for (int d = 0; d < c->world; ++d)
            SCHK(c, hipMemcpyAsync(c->slot(c->rank) + (size_t)d * ch, send + (size_t)d * recvcount + off, m * 8, hipMemcpyDeviceToHost, st));
        SCHK(c, hipStreamSynchronize(st));
        SRC(c, c->barrier());
        SRC(c, c->sum_slots(recv + off, (size_t)c->rank * ch, m));
        SRC(c, c->barrier());
    }
    return GMMIV_OK;
}

int shm_allgather(gmmiv_comm *c, const double *send, double *recv, size_t sendcount)
{
    hipStream_t st = c->ctx->stream;
    const size_t ch = c->slot_elems();
    for (size_t off = 0; off < sendcount; off += ch) {
        const size_t m = sendcount - off < ch ? sendcount - off : ch;
        SCHK(c, hipMemcpyAsync(c->slot(c->rank), send + off, m * 8, hipMemcpyDeviceToHost, st));
        SCHK(c, hipStreamSynchronize(st));
        SRC(c, c->barrier());
        for (int r = 0; r < c->world; ++r)
            SCHK(c, hipMemcpyAsync(recv + (size_t)r * sendcount + off, c->slot(r), m * 8, hipMemcpyHostToDevice, st));
        SCHK(c, hipStreamSynchronize(st));
        SRC(c, c->barrier());
    }
    return GMMIV_OK;
}

int shm_broadcast(gmmiv_comm *c, double *buf, size_t n, int root)
{
    hipStream_t st = c->ctx->stream;
    const size_t ch = c->slot_elems();
    for (size_t off = 0; off < n; off += ch) {
        const size_t m = n - off < ch ? n - off : ch;
        if (c->rank == root) {
            SCHK(c, hipMemcpyAsync(c->slot(root), buf + off, m * 8, hipMemcpyDeviceToHost, st));
            SCHK(c, hipStreamSynchronize(st));
        }
        SRC(c, c->barrier());
        if (c->rank != root) {
            SCHK(c, hipMemcpyAsync(buf + off, c->slot(root), m * 8, hipMemcpyHostToDevice, st));
            SCHK(c, hipStreamSynchronize(st));
        }
        SRC(c, c->barrier());
    }
    return GMMIV_OK;
}

// rank 0 creates and initialises the file named in the id, the others wait for it; a first barrier, then rank 0 unlinks the
// name (the mappings keep the memory alive, nothing is left behind when the ranks exit or crash later)
int shm_attach(gmmiv_comm *c, const char *path)
{
    const char *mb = getenv("GMMIV_COMM_SHM_SLOT_MB");
    size_t slot_bytes = (size_t)(mb && atol(mb) > 0 ? atol(mb) : 16) << 20;
    const size_t total = SHM_HDR_BYTES + slot_bytes * (size_t)c->world;
    int fd = -1;
    if (c->rank == 0) {
        fd = open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { if (fd >= 0) { close(fd); unlink(path); } gmmiv_set_error("gmmiv_comm (shm): cannot create %s (%zu MiB)", path, total >> 20); return GMMIV_ERR_HIP; }
    } else {
        const double t0 = now_s();
        for (;;) {
            fd = open(path, O_RDWR);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= SHM_HDR_BYTES) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (now_s() - t0 > c->timeout_s) { gmmiv_set_error("gmmiv_comm (shm): rank %d waited %.0f s for %s", c->rank, c->timeout_s, path); return GMMIV_ERR_HIP; }
            usleep(2000);
        }
    }
    size_t map_bytes = total;
    if (c->rank != 0) { // the slot size is rank 0's: read it from the header page first
        void *h = mmap(nullptr, SHM_HDR_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (h == MAP_FAILED) { close(fd); gmmiv_set_error("gmmiv_comm (shm): mmap of %s failed", path); return GMMIV_ERR_HIP; }
        ShmHdr *hh = (ShmHdr *)h;
        const double t0 = now_s();
        while (hh->ready.load(std::memory_order_acquire) != 1) {
            if (now_s() - t0 > c->timeout_s) { munmap(h, SHM_HDR_BYTES); close(fd); gmmiv_set_error("gmmiv_comm (shm): rank %d: %s never became ready", c->rank, path); return GMMIV_ERR_HIP; }
            usleep(1000);
        }
        const bool ok = memcmp(hh->magic, SHM_MAGIC, 8) == 0 && hh->world == (uint32_t)c->world;
        map_bytes = SHM_HDR_BYTES + (size_t)hh->slot_bytes * (size_t)c->world;
        munmap(h, SHM_HDR_BYTES);
        if (!ok) { close(fd); gmmiv_set_error("gmmiv_comm (shm): %s belongs to another job (world mismatch)", path); return GMMIV_ERR_ARG; }
    }
    void *m = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { if (c->rank == 0) unlink(path); gmmiv_set_error("gmmiv_comm (shm): mmap of %zu MiB failed", map_bytes >> 20); return GMMIV_ERR_HIP; }
    c->shm = (ShmHdr *)m;
    c->shm_bytes = map_bytes;
    c->shm_path = path;
    if (c->rank == 0) {
        memcpy(c->shm->magic, SHM_MAGIC, 8);
        c->shm->world = (uint32_t)c->world;
        c->shm->slot_bytes = slot_bytes;
        c->shm->count.store(0); c->shm->gen.store(0); c->shm->failed.store(0);
        c->shm->ready.store(1, std::memory_order_release);
    }
    hipError_t e = hipMalloc(&c->addbuf, (size_t)c->shm->slot_bytes);
    if (e != hipSuccess) {
        c->shm->failed.store(1); // peers spinning in the barrier stop waiting
        if (c->rank == 0) unlink(path);
        gmmiv_set_error("gmmiv_comm (shm): hipMalloc of the slot staging failed (%s)", hipGetErrorString(e));
        return GMMIV_ERR_HIP;
    }
    int rc = c->barrier();
    if (c->rank == 0) unlink(path);
    if (rc) return rc;
    char b[256];
    snprintf(b, sizeof(b), "shm (host shared-memory staging, device-side sums; %d ranks, %zu MiB slots)", c->world, (size_t)c->shm->slot_bytes >> 20);
    c->backend = b;
    return GMMIV_OK;
}

// the id files rank 0 published (gmmiv_comm_exchange_id_file): removed once the communicator exists, so that a later job that
// reuses the path cannot pick up this job's id
std::mutex g_idfile_mu;
std::vector<std::pair<std::string, std::string> > g_idfiles; // (id bytes, path)

} // namespace

// Release everything that needs the context (called by gmmiv_comm_destroy, and by gmmiv_ctx_destroy for communicators the
// caller still holds: afterwards the handle is an empty shell that gmmiv_comm_destroy can still delete safely).
void gmmiv_comm_orphan(gmmiv_comm *c)
{
    if (!c || !c->ctx) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->nc) { (void)c->api->CommDestroy(c->nc); c->nc = nullptr; }
    if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
    if (c->addbuf) { (void)hipFree(c->addbuf); c->addbuf = nullptr; }
    if (c->shm) { munmap((void *)c->shm, c->shm_bytes); c->shm = nullptr; }
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); (void)hipStreamDestroy(c->side);
        c->side = nullptr; c->pending = false;
    }
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

int gmmiv_comm_get_unique_id_for(const char *transport, void *id128)
{
    if (!id128) { gmmiv_set_error("comm_get_unique_id: id == NULL"); return GMMIV_ERR_ARG; }
    if (!transport || !*transport) transport = getenv("GMMIV_COMM_TRANSPORT");
    if (!transport || !*transport) transport = "rccl";
    if (!strcmp(transport, "shm")) {
        const char *dir = getenv("GMMIV_COMM_SHM_DIR");
        if (!dir || !*dir) dir = "/dev/shm";
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        static std::atomic<unsigned> serial{0};
        char *id = (char *)id128;
        memset(id, 0, GMMIV_COMM_ID_BYTES);
        memcpy(id, SHM_MAGIC, 8);
        const int n = snprintf(id + 8, GMMIV_COMM_ID_BYTES - 8, "%s/gmmiv_comm_%ld_%lld%09ld_%u", dir, (long)getpid(), (long long)ts.tv_sec, ts.tv_nsec, serial.fetch_add(1));
        if (n <= 0 || n >= GMMIV_COMM_ID_BYTES - 8) { gmmiv_set_error("comm_get_unique_id: GMMIV_COMM_SHM_DIR too long"); return GMMIV_ERR_ARG; }
        return GMMIV_OK;
    }
    if (strcmp(transport, "rccl")) { gmmiv_set_error("comm_get_unique_id: unknown transport \"%s\" (rccl, shm)", transport); return GMMIV_ERR_ARG; }
    const Rccl *api = rccl();
    if (!api) return GMMIV_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == GMMIV_COMM_ID_BYTES, "gmmiv.h and rccl.h disagree on the id size");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { gmmiv_set_error("ncclGetUniqueId -> %s", api->GetErrorString(r)); return GMMIV_ERR_HIP; }
    memcpy(id128, &id, sizeof(id));
    return GMMIV_OK;
}

int gmmiv_comm_get_unique_id(void *id128) { return gmmiv_comm_get_unique_id_for(nullptr, id128); }

// The job's nonce, written behind the id and checked by the readers: a file left at the same path by ANOTHER job (one that died
// between publishing its id and creating its communicator) carries a different nonce and is never accepted, however recent it is.
// GMMIV_COMM_JOB if the launcher sets it, else what torch.distributed.run gives every rank of one job (rendezvous address, port and
// run id); empty when neither exists -- then only the age rule below protects the readers.
// The file carries a FIXED-LENGTH tag of it ("n1:" + 16 hex digits of its 64-bit FNV-1a hash): a job name or rendezvous address of
// any length compares equal on every rank (round 4 wrote the raw string and read at most 512 bytes of it back -- a longer nonce could
// never match and the other ranks spun until the timeout without a word).
static std::string job_nonce_raw()
{
    if (const char *j = getenv("GMMIV_COMM_JOB")) return std::string("job:") + j;
    const char *port = getenv("MASTER_PORT");
    if (!port || !*port) return std::string();
    const char *addr = getenv("MASTER_ADDR"), *run = getenv("TORCHELASTIC_RUN_ID");
    return std::string("rdzv:") + (addr ? addr : "") + ":" + port + ":" + (run ? run : "");
}
static const size_t NONCE_TAG_BYTES = 19;
static std::string job_nonce()
{
    const std::string raw = job_nonce_raw();
    unsigned long long h = 1469598103934665603ULL;
    for (unsigned char ch : raw) { h ^= ch; h *= 1099511628211ULL; }
    char tag[32];
    snprintf(tag, sizeof tag, "n1:%016llx", h);
    return std::string(tag, NONCE_TAG_BYTES);
}

// rank 0: the id file of `id128` is no longer needed (communicator created, or its creation failed)
static void retire_id_file(const void *id128)
{
    if (!id128) return;
    std::lock_guard<std::mutex> lk(g_idfile_mu);
    const std::string key((const char *)id128, GMMIV_COMM_ID_BYTES);
    for (size_t i = 0; i < g_idfiles.size(); ++i)
        if (g_idfiles[i].first == key) { (void)unlink(g_idfiles[i].second.c_str()); g_idfiles.erase(g_idfiles.begin() + i); break; }
}

int gmmiv_comm_exchange_id_file(const char *path, int rank, void *id128, double timeout_s)
{
    if (!path || !id128 || rank < 0) { gmmiv_set_error("comm_exchange_id_file: bad argument"); return GMMIV_ERR_ARG; }
    const std::string nonce = job_nonce();
    if (rank == 0) {
        (void)unlink(path); // a file left by a job that died before its communicator existed
        int rc = gmmiv_comm_get_unique_id(id128);
        if (rc) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id128, 1, GMMIV_COMM_ID_BYTES, f) != GMMIV_COMM_ID_BYTES || fwrite(nonce.data(), 1, nonce.size(), f) != nonce.size()) {
            if (f) fclose(f);
            (void)unlink(tmp.c_str());
            gmmiv_set_error("comm_exchange_id_file: cannot write %s", tmp.c_str());
            return GMMIV_ERR_ARG;
        }
        fclose(f);
        if (rename(tmp.c_str(), path) != 0) { gmmiv_set_error("comm_exchange_id_file: cannot rename %s", tmp.c_str()); return GMMIV_ERR_ARG; }
        std::lock_guard<std::mutex> lk(g_idfile_mu);
        g_idfiles.emplace_back(std::string((const char *)id128, GMMIV_COMM_ID_BYTES), std::string(path));
        return GMMIV_OK;
    }
    // Only a recent file is this job's: one whose modification time lies more than 10 minutes before this call is a leftover of
    // an earlier job at the same path (rank 0 removes its file as soon as the communicator exists, so a leftover means that job
    // died in between) and is ignored until rank 0 replaces it.  The window is wide on purpose: the ranks of one job may reach this
    // call minutes apart (a cold `import torch` on a fresh box takes 1-2 minutes).
    struct timespec t_enter;
    clock_gettime(CLOCK_REALTIME, &t_enter);
    const double step = 0.01;
    int seen_other = 0, seen_stale = 0, seen_short = 0; // why files that WERE there have been passed over (reported on timeout)
    for (double waited = 0.0; waited <= timeout_s; waited += step) { // the rename above makes the file appear complete
        struct stat sb;
        if (stat(path, &sb) == 0) {
            if ((double)sb.st_mtime < (double)t_enter.tv_sec - 600.0) seen_stale = 1;
            else {
                FILE *f = fopen(path, "rb");
                if (f) {
                    char buf[GMMIV_COMM_ID_BYTES + NONCE_TAG_BYTES + 1];
                    const size_t n = fread(buf, 1, sizeof(buf), f);
                    fclose(f);
                    // the file of THIS job: a complete id followed by this job's tag (another job's file is left alone until rank 0 replaces it)
                    if (n == GMMIV_COMM_ID_BYTES + NONCE_TAG_BYTES && memcmp(buf + GMMIV_COMM_ID_BYTES, nonce.data(), NONCE_TAG_BYTES) == 0) {
                        memcpy(id128, buf, GMMIV_COMM_ID_BYTES);
                        return GMMIV_OK;
                    }
                    if (n == GMMIV_COMM_ID_BYTES + NONCE_TAG_BYTES) seen_other = 1;
                    else seen_short = 1; // no tag / another layout: written by another build of the library
                }
            }
        }
        usleep((useconds_t)(step * 1e6));
    }
    gmmiv_set_error("comm_exchange_id_file: rank %d waited %.0f s for %s%s%s%s", rank, timeout_s, path,
                    seen_other ? " -- a file was there, but it carries ANOTHER job's tag (GMMIV_COMM_JOB / MASTER_ADDR:MASTER_PORT:TORCHELASTIC_RUN_ID differ between the ranks?)" : "",
                    seen_short ? " -- a file without this build's job tag was there (rank 0 runs another build of libgmmiv?)" : "",
                    seen_stale ? " -- a file older than 10 minutes was there and was ignored as a leftover" : "");
    return GMMIV_ERR_HIP;
}

int gmmiv_comm_create(gmmiv_ctx *ctx, int world, int rank, const void *id128, gmmiv_comm **out)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) { gmmiv_set_error("comm_create: bad argument"); return GMMIV_ERR_ARG; }
    GCHK(hipSetDevice(ctx->device));
    gmmiv_comm *c = new gmmiv_comm();
    c->ctx = ctx; c->world = world; c->rank = rank;
    if (const char *t = getenv("GMMIV_COMM_TIMEOUT_S")) { if (atof(t) > 0) c->timeout_s = atof(t); }
    if (world > 1 && id128 && memcmp(id128, SHM_MAGIC, 8) == 0) {
        char path[GMMIV_COMM_ID_BYTES - 8 + 1];
        memcpy(path, (const char *)id128 + 8, GMMIV_COMM_ID_BYTES - 8);
        path[GMMIV_COMM_ID_BYTES - 8] = 0;
        int rc = shm_attach(c, path);
        if (rc) {
            if (c->addbuf) (void)hipFree(c->addbuf);
            if (c->shm) munmap((void *)c->shm, c->shm_bytes);
            delete c;
            if (rank == 0) retire_id_file(id128); // a retry must not find this job's dead id
            return rc;
        }
    } else if (world > 1 || (getenv("GMMIV_COMM_FORCE_RCCL") && *getenv("GMMIV_COMM_FORCE_RCCL") == '1')) {
        // (GMMIV_COMM_FORCE_RCCL=1: a ONE-rank communicator goes through RCCL as well -- every entry point below then executes its
        //  RCCL call, side stream and events included, on a one-GPU machine: the check that the dlopen'ed symbols, data types and
        //  stream arguments are right before a multi-GPU node ever sees them)
        c->api = rccl();
        if (!c->api) { delete c; if (rank == 0) retire_id_file(id128); return GMMIV_ERR_UNSUPPORTED; }
        ncclUniqueId id;
        if (id128) memcpy(&id, id128, sizeof(id));
        else { // one forced rank without an id: draw it here
            ncclResult_t gr = c->api->GetUniqueId(&id);
            if (gr != ncclSuccess) { gmmiv_set_error("ncclGetUniqueId -> %s", c->api->GetErrorString(gr)); delete c; return GMMIV_ERR_HIP; }
        }
        ncclResult_t r = c->api->CommInitRank(&c->nc, world, id, rank);
        if (r != ncclSuccess) {
            gmmiv_set_error("ncclCommInitRank(world %d, rank %d, device %d) -> %s", world, rank, ctx->device, c->api->GetErrorString(r));
            delete c;
            if (rank == 0) retire_id_file(id128);
            return GMMIV_ERR_HIP;
        }
        c->backend = "rccl: " + c->api->where;
    }
    if (world > 1 && rank == 0) retire_id_file(id128); // every rank has read the id by now (both transports meet inside the create)
    ctx->comms.push_back(c);
    *out = c;
    return GMMIV_OK;
}

void gmmiv_comm_destroy(gmmiv_comm *c)
{
    if (!c) return;
    gmmiv_comm_orphan(c); // releases RCCL / the shared mapping and the staging buffers (no-op when the context already went away)
    delete c;
}

int gmmiv_comm_world(const gmmiv_comm *c) { return c ? c->world : 0; }
int gmmiv_comm_rank(const gmmiv_comm *c) { return c ? c->rank : -1; }
const char *gmmiv_comm_backend(const gmmiv_comm *c)
{
    if (!c) return "";
    return c->local() ? "single rank (no collective library)" : c->backend.c_str();
}
int gmmiv_comm_info(const gmmiv_comm *c, int *rccl_version, int *rccl_comm_count)
{
    if (!c) { gmmiv_set_error("comm_info: NULL communicator"); return GMMIV_ERR_ARG; }
    if (rccl_version) *rccl_version = 0;
    if (rccl_comm_count) *rccl_comm_count = 0;
    if (!c->nc || !c->api) return GMMIV_OK; // single rank / shm transport: no collective library behind this communicator
    if (rccl_version && c->api->GetVersion) (void)c->api->GetVersion(rccl_version);
    if (rccl_comm_count && c->api->CommCount) (void)c->api->CommCount(c->nc, rccl_comm_count);
    return GMMIV_OK;
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
    if (c->local() || n == 0) return GMMIV_OK;
    GCHK(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    if (gmmiv_is_device_ptr(buf)) {
        if (c->is_shm()) return shm_allreduce(c, buf, n);
        NCHK(c, c->api->AllReduce(buf, buf, n, ncclFloat64, ncclSum, c->nc, st));
        return GMMIV_OK;
    }
    void *d;
    int rc = c->staged(n * 8, &d);
    if (rc) { if (c->is_shm()) c->shm->failed.store(1); return rc; }
    GCHK(hipMemcpyAsync(d, buf, n * 8, hipMemcpyHostToDevice, st));
    if (c->is_shm()) { rc = shm_allreduce(c, (double *)d, n); if (rc) return rc; }
    else NCHK(c, c->api->AllReduce(d, d, n, ncclFloat64, ncclSum, c->nc, st));
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
    if (c->local()) {
        if (recv != send) GCHK(hipMemcpyAsync(recv, send, recvcount * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
        return GMMIV_OK;
    }
    if (c->is_shm()) return shm_reduce_scatter(c, send, recv, recvcount);
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
    if (c->local()) {
        if (recv != send) GCHK(hipMemcpyAsync(recv, send, sendcount * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
        return GMMIV_OK;
    }
    if (c->is_shm()) return shm_allgather(c, send, recv, sendcount);
    NCHK(c, c->api->AllGather(send, recv, sendcount, ncclFloat64, c->nc, c->ctx->stream));
    return GMMIV_OK;
}

// ---- overlapped forms: the collective runs on the communicator's side stream, behind everything enqueued on the context's
// stream so far; gmmiv_comm_join orders the context's stream behind it again.  One rank / shm transport: the plain call.
int gmmiv_allreduce_f64_begin(gmmiv_comm *c, double *buf, size_t n)
{
    if (!c || !c->ctx || (!buf && n)) { gmmiv_set_error("allreduce_f64_begin: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    if (c->local() || n == 0 || c->is_shm()) return gmmiv_allreduce_f64(c, buf, n);
    if (!gmmiv_is_device_ptr(buf)) { gmmiv_set_error("allreduce_f64_begin: device buffers only"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)n * 8;
    GCHK(hipSetDevice(c->ctx->device));
    int rc = c->fork_side();
    if (rc) return rc;
    NCHK(c, c->api->AllReduce(buf, buf, n, ncclFloat64, ncclSum, c->nc, c->side));
    return c->mark_side();
}

int gmmiv_reduce_scatter_f64_begin(gmmiv_comm *c, const double *send, double *recv, size_t recvcount)
{
    if (!c || !c->ctx || ((!send || !recv) && recvcount)) { gmmiv_set_error("reduce_scatter_f64_begin: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    if (c->local() || recvcount == 0 || c->is_shm()) return gmmiv_reduce_scatter_f64(c, send, recv, recvcount);
    if (!gmmiv_is_device_ptr(send) || !gmmiv_is_device_ptr(recv)) { gmmiv_set_error("reduce_scatter_f64_begin: device buffers only"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)recvcount * 8 * c->world;
    GCHK(hipSetDevice(c->ctx->device));
    int rc = c->fork_side();
    if (rc) return rc;
    NCHK(c, c->api->ReduceScatter(send, recv, recvcount, ncclFloat64, ncclSum, c->nc, c->side));
    return c->mark_side();
}

int gmmiv_allgather_f64_begin(gmmiv_comm *c, const double *send, double *recv, size_t sendcount)
{
    if (!c || !c->ctx || ((!send || !recv) && sendcount)) { gmmiv_set_error("allgather_f64_begin: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    if (c->local() || sendcount == 0 || c->is_shm()) return gmmiv_allgather_f64(c, send, recv, sendcount);
    if (!gmmiv_is_device_ptr(send) || !gmmiv_is_device_ptr(recv)) { gmmiv_set_error("allgather_f64_begin: device buffers only"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)sendcount * 8 * c->world;
    GCHK(hipSetDevice(c->ctx->device));
    int rc = c->fork_side();
    if (rc) return rc;
    NCHK(c, c->api->AllGather(send, recv, sendcount, ncclFloat64, c->nc, c->side));
    return c->mark_side();
}

int gmmiv_comm_join(gmmiv_comm *c)
{
    if (!c || !c->ctx) { gmmiv_set_error("comm_join: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    if (!c->pending) return GMMIV_OK;
    GCHK(hipSetDevice(c->ctx->device));
    GCHK(hipStreamWaitEvent(c->ctx->stream, c->ev_join, 0));
    c->pending = false;
    return GMMIV_OK;
}

// buf[n] on every rank <- buf of `root`; host or device
int gmmiv_broadcast_f64(gmmiv_comm *c, double *buf, size_t n, int root)
{
    if (!c || !c->ctx || (!buf && n) || root < 0 || root >= c->world) { gmmiv_set_error("broadcast_f64: bad argument (or the context was destroyed)"); return GMMIV_ERR_ARG; }
    c->bytes_moved += (double)n * 8;
    if (c->local() || n == 0) return GMMIV_OK;
    GCHK(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    if (gmmiv_is_device_ptr(buf)) {
        if (c->is_shm()) return shm_broadcast(c, buf, n, root);
        NCHK(c, c->api->Broadcast(buf, buf, n, ncclFloat64, root, c->nc, st));
        return GMMIV_OK;
    }
    void *d;
    int rc = c->staged(n * 8, &d);
    if (rc) { if (c->is_shm()) c->shm->failed.store(1); return rc; }
    if (c->rank == root) GCHK(hipMemcpyAsync(d, buf, n * 8, hipMemcpyHostToDevice, st));
    if (c->is_shm()) { rc = shm_broadcast(c, (double *)d, n, root); if (rc) return rc; }
    else NCHK(c, c->api->Broadcast(d, d, n, ncclFloat64, root, c->nc, st));
    GCHK(hipMemcpyAsync(buf, d, n * 8, hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    return GMMIV_OK;
}

} // extern "C"
