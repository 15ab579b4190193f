// Context lifecycle, error reporting and device-memory helpers of libpsk.so.
#include "psk_internal.h"

#include <chrono>

#include <mutex>

static thread_local std::string g_init_error;

static std::mutex g_err_mu;

int psk_fail(psk_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);   // (the stages of a call's .gz pipeline run in threads of their own and may fail side by side)
    if (ctx) ctx->err = buf;
    else g_init_error = buf;
    return code;
}

std::string psk_error_text(psk_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    return ctx ? ctx->err : g_init_error;
}

void psk_set_error_text(psk_ctx *ctx, const std::string &text)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    if (ctx) ctx->err = text;
}

int dev_reserve(psk_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return PSK_OK;
    if (b.p && !b.borrowed) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0; b.borrowed = false;
    size_t want = bytes + bytes / 8 + 256;  // slack so that slowly growing samples do not realloc each time
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes ? bytes : 256;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return psk_fail(ctx, PSK_ENOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return PSK_OK;
}

void dev_release(DevBuf &b)
{
    if (b.p && !b.borrowed) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    b.borrowed = false;
}

int arena_alloc(psk_ctx *ctx, size_t bytes, void **out)
{
    Arena &A = ctx->arena;
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    while (A.cur < A.chunks.size()) {  // first chunk, from the current one on, with room
        if (A.off + bytes <= A.sizes[A.cur]) {
            *out = static_cast<uint8_t *>(A.chunks[A.cur]) + A.off;
            A.off += bytes;
            return PSK_OK;
        }
        A.cur++;
        A.off = 0;
    }
    // (r06: the first chunks are 1 GiB, those of a run that has already taken eight of them 4 GiB: config 3 on one GPU holds 122 GB of
    // lists -- 118 hipMallocs while it counts and 118 hipFrees at its end, 20-90 ms each way)
    const size_t step = A.chunks.size() < 8 ? Arena::CHUNK : 4 * Arena::CHUNK;
    const size_t sz = bytes > step ? bytes : step;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) return psk_fail(ctx, PSK_ENOMEM, "hipMalloc(%zu bytes) failed: %s", sz, hipGetErrorString(e));
    A.chunks.push_back(p);
    A.sizes.push_back(sz);
    A.cur = A.chunks.size() - 1;
    A.off = bytes;
    *out = p;
    return PSK_OK;
}

namespace {
struct PinnedCache {
    struct Entry { void *p; size_t cap; int device; };
    std::mutex mu;
    std::vector<Entry> held;
    size_t bytes = 0;
};
PinnedCache g_pinned;
size_t pinned_cache_limit()
{
    const char *e = getenv("PSK_PINNED_CACHE_MB");
    return (size_t)(e && *e ? strtoull(e, nullptr, 10) : 1024) << 20;
}
}  // namespace

int pinned_acquire(psk_ctx *ctx, size_t need, void **buf, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        size_t best = g_pinned.held.size();
        for (size_t i = 0; i < g_pinned.held.size(); i++) {   // the smallest cached buffer of this device that is large enough (and not four times too large)
            const PinnedCache::Entry &e = g_pinned.held[i];
            if (e.device != ctx->device || e.cap < need || e.cap / 4 > need) continue;
            if (best == g_pinned.held.size() || e.cap < g_pinned.held[best].cap) best = i;
        }
        if (best < g_pinned.held.size()) {
            *buf = g_pinned.held[best].p;
            *cap = g_pinned.held[best].cap;
            g_pinned.bytes -= g_pinned.held[best].cap;
            g_pinned.held.erase(g_pinned.held.begin() + (long)best);
            return PSK_OK;
        }
    }
    const size_t want = need + need / 4;
    hipError_t e = hipHostMalloc(buf, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        *buf = nullptr;
        *cap = 0;
        (void)hipGetLastError();
        return psk_fail(ctx, PSK_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    *cap = want;
    return PSK_OK;
}

void pinned_release(psk_ctx *ctx, void *buf, size_t cap)
{
    if (!buf) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        if (cap && g_pinned.bytes + cap <= pinned_cache_limit()) {
            g_pinned.held.push_back({buf, cap, ctx->device});
            g_pinned.bytes += cap;
            return;
        }
    }
    (void)hipHostFree(buf);
}

void arena_release(psk_ctx *ctx)
{
    for (void *p : ctx->arena.chunks) (void)hipFree(p);
    ctx->arena = Arena();
}

void reset_lists(psk_ctx *ctx, int n_samples)
{
    // list storage lives in the arena: rewind it (chunks are kept for the next run)
    ctx->arena.cur = 0;
    ctx->arena.off = 0;
    ctx->lists.assign((size_t)(n_samples > 0 ? n_samples : 0), SampleList());
}

extern "C" int psk_version(void) { return (1 << 16) | 0; }

extern "C" const char *psk_last_error(const psk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

extern "C" int psk_init(int device, psk_ctx **ctx_out)
{
    if (!ctx_out) return PSK_EINVAL;
    *ctx_out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return psk_fail(nullptr, PSK_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return psk_fail(nullptr, PSK_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return psk_fail(nullptr, PSK_EHIP, "hipSetDevice failed: %s", hipGetErrorString(e));
    psk_ctx *ctx = new (std::nothrow) psk_ctx();
    if (!ctx) return psk_fail(nullptr, PSK_ENOMEM, "out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return psk_fail(nullptr, PSK_EHIP, "stream/event creation failed");
    }
    *ctx_out = ctx;
    return PSK_OK;
}

extern "C" void psk_free(psk_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    const bool trace = getenv("PSK_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[psk] psk_free: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    comm_release(ctx);
    reset_lists(ctx, 0);
    const size_t n_chunks = ctx->arena.chunks.size();
    arena_release(ctx);
    if (trace) fprintf(stderr, "[psk] psk_free: %zu arena chunks\n", n_chunks);
    lap("arena released");
    DevBuf *bufs[] = {&ctx->raw, &ctx->keysA, &ctx->keysB, &ctx->valsA, &ctx->valsB, &ctx->hist, &ctx->scan_tmp, &ctx->flags, &ctx->starts,
                      &ctx->misc, &ctx->union_words, &ctx->bits, &ctx->mask1, &ctx->phe,
                      &ctx->slot[0].res, &ctx->slot[1].res, &ctx->res_count, &ctx->res_sorted, &ctx->lut, &ctx->bs_spl, &ctx->bs_ct};
    for (DevBuf *b : bufs) dev_release(*b);
    lap("matrix, union, scan buffers released");
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (CountLane &L : ctx->lane) {
        DevBuf *lb[] = {&L.raw, &L.keysA, &L.keysB, &L.starts, &L.cnt, &L.sk_cand, &L.sk_out,
                        &L.dc_part, &L.dc_wgoff, &L.dc_cnt, &L.dc_meta, &L.dc_mtemp, &L.rawin, &L.fr_scratch};
        for (DevBuf *b : lb) dev_release(*b);
        if (L.pinned_cnt && !(ctx->lane_pinned && L.pinned_cnt >= ctx->lane_pinned && L.pinned_cnt < ctx->lane_pinned + 16 * psk_ctx::LANES))
            (void)hipHostFree(L.pinned_cnt);
        if (L.sk_host) (void)hipHostFree(L.sk_host);
        if (L.sk_done) (void)hipEventDestroy(L.sk_done);
        if (L.sk_filtered) (void)hipEventDestroy(L.sk_filtered);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.raw_ready) (void)hipEventDestroy(L.raw_ready);
        if (L.raw_free) (void)hipEventDestroy(L.raw_free);
        if (L.up_done) (void)hipEventDestroy(L.up_done);
    }
    lap("lane buffers and events released");
    gz_release(ctx);
    if (ctx->gz_stream) (void)hipStreamDestroy(ctx->gz_stream);
    if (ctx->gz_up_stream) (void)hipStreamDestroy(ctx->gz_up_stream);
    if (trace) fprintf(stderr, "[psk] psk_free: lane slab of %.2f GB\n", ctx->lane_slab.cap / 1e9);
    dev_release(ctx->lane_slab);
    lap("lane slab released");
    if (ctx->lane_pinned) (void)hipHostFree(ctx->lane_pinned);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipStream_t &cs : ctx->copy_more) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); cs = nullptr; }
    if (ctx->frame_stream) { (void)hipStreamSynchronize(ctx->frame_stream); (void)hipStreamDestroy(ctx->frame_stream); }
    if (ctx->sketch_stream) { (void)hipStreamSynchronize(ctx->sketch_stream); (void)hipStreamDestroy(ctx->sketch_stream); }
    lap("gz buffers, lane slab, streams released");
    pinned_release(ctx, ctx->pinned, ctx->pinned_cap);
    if (ctx->scan_pinned) (void)hipHostFree(ctx->scan_pinned);
    if (ctx->cnt_pinned) (void)hipHostFree(ctx->cnt_pinned);
    for (size_t r = 0; r < ctx->ring.size(); r++) pinned_release(ctx, ctx->ring[r], r < ctx->ring_cap.size() ? ctx->ring_cap[r] : 0);   // (into the process's cache: psk_internal.h)
    if (trace) fprintf(stderr, "[psk] psk_free: %zu ring slots\n", ctx->ring.size());
    lap("pinned ring released");
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (ScanSlot &sl : ctx->slot)
        for (hipEvent_t e : {sl.ev0, sl.ev1, sl.ev_export}) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    lap("events, main stream released");
    delete ctx;
}

extern "C" int psk_device_info(psk_ctx *ctx, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes)
{
    if (!ctx) return PSK_EINVAL;
    hipDeviceProp_t prop;
    PSK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap > 0) {
        // gcnArchName tells gfx950 apart; name is the marketing string
        snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return PSK_OK;
}

extern "C" int psk_begin(psk_ctx *ctx, int k, int n_samples, uint64_t slab_lo, uint64_t slab_hi)
{
    if (!ctx) return PSK_EINVAL;
    if (k < 1 || k > 32) return psk_fail(ctx, PSK_EINVAL, "k-mer length must be 1..32, got %d", k);
    if (n_samples < 1) return psk_fail(ctx, PSK_EINVAL, "n_samples must be >= 1");
    if (slab_hi != 0 && slab_hi <= slab_lo) return psk_fail(ctx, PSK_EINVAL, "empty slab");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    // a new run: nothing of the previous one may still be in flight, and its scan results are gone
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (ScanSlot &sl : ctx->slot) {
        if (sl.export_pending) PSK_HIP(ctx, hipEventSynchronize(sl.ev_export));  // an export on the caller's stream
        sl.in_flight = false;
        sl.export_pending = false;
    }
    ctx->n_in_flight = 0;
    ctx->results_valid = false;
    ctx->dense_hint = -1;
    reset_lists(ctx, n_samples);
    ctx->k = k;
    ctx->n_samples = n_samples;
    ctx->slab_lo = slab_lo;
    ctx->slab_hi = slab_hi;
    dense_configure(ctx);
    ctx->bs_ready = false;   // the splitters of the bucketed sort belong to a run (its k, its slab)
    for (CountLane &L : ctx->lane) L.dc_slot = 0;   // the dense and the bucketed route lay the counter ring out differently: zero it again
    // a large slab of a grouped batch (long samples) does not stay beside the lists and the matrix of the next run
    if (ctx->lane_slab.cap > ((size_t)4 << 30)) {
        psk_forget_lane_slices(ctx);
        dev_release(ctx->lane_slab);
    }
    ctx->n_kmers = 0;
    ctx->have_presence = false;
    ctx->last = ScanParams();
    ctx->last_scan_kind = 0;
    ctx->n_pass = 0;
    return PSK_OK;
}
