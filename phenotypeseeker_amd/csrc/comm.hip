// e: the collectives of the range-sharded path, on RCCL over xGMI, bound directly (no PyTorch): one communicator per
// context, its own HIP stream, librccl.so opened on first use so that single-GPU runs never load it.
//   all-reduce   union sizes -> global Bonferroni denominator (modeling.py:644,:738,:795), timings, barriers
//   all-gather   the survivors of every slab after a scan (device buffers, asynchronous on the comm stream)
//   all-to-all   slab ranges of the sorted per-sample lists (multi-GPU ingest), ncclSend/ncclRecv in one group
// The unique id of ncclGetUniqueId travels between the ranks through the caller's rendezvous (a file or a
// socket; phenotypeseeker_amd/dist.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

#include "psk_internal.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl(psk_ctx *ctx)
{
    if (g_rccl.lib) return PSK_OK;
    const char *names[] = {getenv("PSK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return psk_fail(ctx, PSK_ESTATE, "cannot open librccl.so (%s): multi-GPU runs need RCCL", dlerror());
    RcclApi a;
    a.lib = lib;
#define PSK_SYM(field, name)                                                                    \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));                            \
    if (!a.field) { dlclose(lib); return psk_fail(ctx, PSK_ESTATE, "librccl.so lacks %s", name); }
    PSK_SYM(GetUniqueId, "ncclGetUniqueId")
    PSK_SYM(CommInitRank, "ncclCommInitRank")
    PSK_SYM(CommDestroy, "ncclCommDestroy")
    PSK_SYM(CommCount, "ncclCommCount")
    PSK_SYM(AllReduce, "ncclAllReduce")
    PSK_SYM(AllGather, "ncclAllGather")
    PSK_SYM(Send, "ncclSend")
    PSK_SYM(Recv, "ncclRecv")
    PSK_SYM(GroupStart, "ncclGroupStart")
    PSK_SYM(GroupEnd, "ncclGroupEnd")
    PSK_SYM(GetErrorString, "ncclGetErrorString")
#undef PSK_SYM
    g_rccl = a;
    return PSK_OK;
}

}  // namespace

struct PskComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    DevBuf stage_a, stage_b;   // device staging of the host-buffer collectives
};

#define PSK_NCCL(ctx, call)                                                                          \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess)                                                                       \
            return psk_fail((ctx), PSK_EHIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), \
                            __FILE__, __LINE__);                                                     \
    } while (0)

void comm_release(psk_ctx *ctx)
{
    PskComm *c = static_cast<PskComm *>(ctx->comm);
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    dev_release(c->stage_a);
    dev_release(c->stage_b);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
}

static PskComm *comm_of(psk_ctx *ctx)
{
    return ctx ? static_cast<PskComm *>(ctx->comm) : nullptr;
}

extern "C" int psk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int psk_comm_unique_id(psk_ctx *ctx, uint8_t *id_out, int cap)
{
    if (!id_out || cap < (int)sizeof(ncclUniqueId))
        return psk_fail(ctx, PSK_ERANGE, "the unique id needs %d bytes", (int)sizeof(ncclUniqueId));
    PSK_TRY(load_rccl(ctx));
    ncclUniqueId id;
    PSK_NCCL(ctx, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return (int)sizeof id;
}

extern "C" int psk_comm_init(psk_ctx *ctx, const uint8_t *id, int id_len, int rank, int world)
{
    if (!ctx) return PSK_EINVAL;
    if (!id || id_len != (int)sizeof(ncclUniqueId)) return psk_fail(ctx, PSK_EINVAL, "bad unique id");
    if (world < 1 || rank < 0 || rank >= world) return psk_fail(ctx, PSK_EINVAL, "rank %d outside a world of %d", rank, world);
    if (ctx->comm) return psk_fail(ctx, PSK_ESTATE, "the context already has a communicator");
    PSK_TRY(load_rccl(ctx));
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PskComm *c = new (std::nothrow) PskComm();
    if (!c) return psk_fail(ctx, PSK_ENOMEM, "out of host memory");
    c->rank = rank;
    c->world = world;
    ctx->comm = c;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        comm_release(ctx);
        return psk_fail(ctx, PSK_EHIP, "stream creation failed");
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        comm_release(ctx);
        return psk_fail(ctx, PSK_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    return PSK_OK;
}

// Ranks of the communicator as RCCL itself counts them (ncclCommCount): what a multi-GPU measurement quotes as
// proof that its collectives ran on RCCL with every rank joined.  < 0 on error (no communicator: PSK_ESTATE).
extern "C" int psk_comm_size(psk_ctx *ctx)
{
    PskComm *c = comm_of(ctx);
    if (!c || !c->comm) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    int n = 0;
    PSK_NCCL(ctx, g_rccl.CommCount(c->comm, &n));
    return n;
}

extern "C" int psk_comm_free(psk_ctx *ctx)
{
    if (!ctx) return PSK_EINVAL;
    comm_release(ctx);
    return PSK_OK;
}

extern "C" void *psk_comm_stream(psk_ctx *ctx)
{
    PskComm *c = comm_of(ctx);
    return c ? static_cast<void *>(c->stream) : nullptr;
}

extern "C" int psk_comm_sync(psk_ctx *ctx)
{
    PskComm *c = comm_of(ctx);
    if (!c) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    PSK_HIP(ctx, hipStreamSynchronize(c->stream));
    return PSK_OK;
}

// dtype: 0 = u64, 1 = f64; op: 0 = sum, 1 = max.  `vals` (host) is reduced in place over the ranks.
extern "C" int psk_comm_allreduce(psk_ctx *ctx, void *vals, int count, int dtype, int op)
{
    PskComm *c = comm_of(ctx);
    if (!c) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    if (!vals || count < 1 || dtype < 0 || dtype > 1 || op < 0 || op > 1) return psk_fail(ctx, PSK_EINVAL, "bad all-reduce arguments");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)count * 8;
    PSK_TRY(dev_reserve(ctx, c->stage_a, bytes));
    PSK_HIP(ctx, hipMemcpyAsync(c->stage_a.p, vals, bytes, hipMemcpyHostToDevice, c->stream));
    PSK_NCCL(ctx, g_rccl.AllReduce(c->stage_a.p, c->stage_a.p, (size_t)count, dtype ? ncclFloat64 : ncclUint64,
                                   op ? ncclMax : ncclSum, c->comm, c->stream));
    PSK_HIP(ctx, hipMemcpyAsync(vals, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
    PSK_HIP(ctx, hipStreamSynchronize(c->stream));
    return PSK_OK;
}

// Host buffers: recv[world][bytes] <- every rank's send[bytes]; staged through device memory, waited for.
extern "C" int psk_comm_allgather_host(psk_ctx *ctx, const void *send, void *recv, uint64_t bytes)
{
    PskComm *c = comm_of(ctx);
    if (!c) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    if (bytes == 0) return PSK_OK;
    if (!send || !recv) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_TRY(dev_reserve(ctx, c->stage_a, bytes));
    PSK_TRY(dev_reserve(ctx, c->stage_b, bytes * (uint64_t)c->world));
    PSK_HIP(ctx, hipMemcpyAsync(c->stage_a.p, send, bytes, hipMemcpyHostToDevice, c->stream));
    PSK_NCCL(ctx, g_rccl.AllGather(c->stage_a.p, c->stage_b.p, bytes, ncclUint8, c->comm, c->stream));
    PSK_HIP(ctx, hipMemcpyAsync(recv, c->stage_b.p, bytes * (uint64_t)c->world, hipMemcpyDeviceToHost, c->stream));
    PSK_HIP(ctx, hipStreamSynchronize(c->stream));
    return PSK_OK;
}

// Device buffers, queued on the communicator's stream and NOT waited for (psk_comm_sync, or a later copy on
// that stream, orders behind it): the survivor exchange after a scan.
extern "C" int psk_comm_allgather_device(psk_ctx *ctx, const void *send_dev, void *recv_dev, uint64_t bytes)
{
    PskComm *c = comm_of(ctx);
    if (!c) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    if (bytes == 0) return PSK_OK;
    if (!send_dev || !recv_dev) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_NCCL(ctx, g_rccl.AllGather(send_dev, recv_dev, bytes, ncclUint8, c->comm, c->stream));
    return PSK_OK;
}

// all-to-all(v) of device buffers: send_counts[d] elements of elem_bytes (4 or 8) go to rank d, taken back to
// back from send_dev; recv_counts[s] elements arrive from rank s, stored back to back in recv_dev.  Waited for.
extern "C" int psk_comm_alltoallv_device(psk_ctx *ctx, const void *send_dev, const uint64_t *send_counts, void *recv_dev,
                                         const uint64_t *recv_counts, int elem_bytes)
{
    PskComm *c = comm_of(ctx);
    if (!c) return psk_fail(ctx, PSK_ESTATE, "no communicator");
    if (!send_counts || !recv_counts || (elem_bytes != 4 && elem_bytes != 8)) return psk_fail(ctx, PSK_EINVAL, "bad all-to-all arguments");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const ncclDataType_t dt = elem_bytes == 8 ? ncclUint64 : ncclUint32;
    const uint8_t *sp = static_cast<const uint8_t *>(send_dev);
    uint8_t *rp = static_cast<uint8_t *>(recv_dev);
    uint64_t tot_s = 0, tot_r = 0;
    for (int r = 0; r < c->world; r++) { tot_s += send_counts[r]; tot_r += recv_counts[r]; }
    if ((tot_s && !send_dev) || (tot_r && !recv_dev)) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_NCCL(ctx, g_rccl.GroupStart());
    // an error between GroupStart and GroupEnd must not leave the group open: remember it, close the group, report it
    ncclResult_t bad = ncclSuccess;
    // a message is cut into pieces of at most 2^27 elements (1 GiB of words): one ncclSend of 10.7 GB -- a rank's share of config 5
    // sent to itself on a one-rank communicator -- arrived damaged (r05: the receiving context found lists that did not ascend);
    // sends and receives between two ranks are matched in the order they are posted, so the pieces pair up
    const uint64_t piece = 1ull << 27;
    for (int r = 0; r < c->world && bad == ncclSuccess; r++) {
        for (uint64_t off = 0; off < send_counts[r] && bad == ncclSuccess; off += piece)
            bad = g_rccl.Send(sp + off * (uint64_t)elem_bytes, (size_t)std::min(piece, send_counts[r] - off), dt, r, c->comm, c->stream);
        for (uint64_t off = 0; off < recv_counts[r] && bad == ncclSuccess; off += piece)
            bad = g_rccl.Recv(rp + off * (uint64_t)elem_bytes, (size_t)std::min(piece, recv_counts[r] - off), dt, r, c->comm, c->stream);
        sp += send_counts[r] * (uint64_t)elem_bytes;
        rp += recv_counts[r] * (uint64_t)elem_bytes;
    }
    const ncclResult_t ended = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return psk_fail(ctx, PSK_EHIP, "ncclSend / ncclRecv failed inside the all-to-all group: %s", g_rccl.GetErrorString(bad));
    PSK_NCCL(ctx, ended);
    PSK_HIP(ctx, hipStreamSynchronize(c->stream));
    return PSK_OK;
}

// ---- plain device buffers for the callers of the exchanges (send / receive buffers live outside the context) ---
extern "C" int psk_dev_alloc(psk_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return PSK_EINVAL;
    *out = nullptr;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 256);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return psk_fail(ctx, PSK_ENOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    return PSK_OK;
}

extern "C" int psk_dev_free(psk_ctx *ctx, void *p)
{
    if (!ctx) return PSK_EINVAL;
    if (!p) return PSK_OK;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_HIP(ctx, hipFree(p));
    return PSK_OK;
}

// kind: 0 host -> device, 1 device -> host, 2 device -> device; on_comm_stream != 0 orders the copy behind the
// collectives queued on the communicator's stream.  Waited for.
extern "C" int psk_dev_copy(psk_ctx *ctx, void *dst, const void *src, uint64_t bytes, int kind, int on_comm_stream)
{
    if (!ctx) return PSK_EINVAL;
    if (bytes == 0) return PSK_OK;
    if (!dst || !src || kind < 0 || kind > 2) return psk_fail(ctx, PSK_EINVAL, "bad copy arguments");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PskComm *c = comm_of(ctx);
    hipStream_t st = (on_comm_stream && c) ? c->stream : ctx->stream;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    PSK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, k, st));
    PSK_HIP(ctx, hipStreamSynchronize(st));
    return PSK_OK;
}
