// a2 + a3 for k >= 14 (config 3: 2,048 samples, k = 16): union + presence matrix by a STREAMING MERGE of the
// per-sample lists, which are already sorted -- no pair sort (get_feature_vector / get_union / map_samples,
// modeling.py:317-380: the reference merges the lists on disk with glistcompare -u and maps every sample back with
// glistquery -l).
//
// r02's route packed every (word, sample) pair into a u64 and radix-sorted 1.28 G of them per slab: 111 ms and
// ~424 GB of HBM traffic for 21 GB of algorithmic bytes.  Here the slab's word range is cut into tiles at
// quantiles of a pilot of the lists (~8,192 pairs each; a tile spans at most PM_BMW * 64 word values), consecutive
// tiles form a range, and ONE workgroup streams a range for a group of 1,024 samples: lane l of wave v owns sample
// 1024 g + 64 v + l, finds its cursor once (one binary search per lane per workgroup) and from then on only moves
// forward -- the end of a tile is the start of the next, so there is no tile table at all.  Inside a tile a wave runs a
// 64-way merge: m = minimum of the lanes' current words (DPP min-reduce), ballot(current == m) IS the 64 presence bits
// of word m for the wave's samples, the lanes that hit advance.  Words shared by many samples (the ancestral k-mers:
// 95 % of the pairs of config 3) cost one iteration per wave instead of 64 atomics.
//   pass 1  pm_mark:   per tile an LDS bitmap of the word values that occur -> global occupancy bitmap (1 bit per
//                      word value of the slab: 35-190 MB at config 3)
//   (scan)  popcounts of the bitmap words -> exclusive scan = rank of every word value = its row; M = total
//   pass 2  pm_fill:   the same stream again; row of m = rank[(m - lo) >> 6] + popcount of the lower bits; wave v
//                      stores its ballot into column v of the tile's LDS block (every (row, column) is written at
//                      most once: plain stores, no atomics); rows are copied out coalesced, 128 B per row and group;
//                      group 0 also expands the bitmap into the union words
// Traffic: lists read twice (2 x 8 B per pair) + matrix written once + the bitmap; no (word, sample) pair is ever
// written.  Rows come out in ascending word order (glistcompare's).
#include "dev_utils.h"
#include "psk_internal.h"

#include <algorithm>
#include <chrono>

namespace {

constexpr int PM_GROUP = 1024;          // samples per workgroup: 16 waves x 64 lanes
constexpr int PM_COLS = PM_GROUP / 64;  // u64 columns of a row one group writes
constexpr uint32_t PM_BMW = 2048;       // most bitmap words (64 word values each) a tile may span: 16 KB + 8 KB of ranks
constexpr uint64_t PM_SENT = ~0ull;     // "no word": beyond any canonical word of an eligible run (2k <= 34)
constexpr size_t PM_LDS_MAX = 144 * 1024;

struct PmList {
    const uint64_t *words;
    uint64_t n;
};

__device__ __forceinline__ uint64_t pm_dpp_u64(uint64_t v, const int tag)
{
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    switch (tag) {  // constant-folded (the builtin wants an immediate control word)
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
    }
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// minimum over the 64 lanes, wave-uniform; all lanes active.  Quad steps, then the mirrors act as xor-4 / xor-8
// butterflies on values that are already uniform per quad / per 8; the four rows meet through scalar lane reads.
__device__ __forceinline__ uint64_t pm_wave_min(uint64_t v)
{
    uint64_t o;
    o = pm_dpp_u64(v, 0); v = o < v ? o : v;
    o = pm_dpp_u64(v, 1); v = o < v ? o : v;
    o = pm_dpp_u64(v, 2); v = o < v ? o : v;
    o = pm_dpp_u64(v, 3); v = o < v ? o : v;
    const uint64_t a = psk_readlane_u64(v, 0), b = psk_readlane_u64(v, 16), c = psk_readlane_u64(v, 32), d = psk_readlane_u64(v, 48);
    const uint64_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// A lane's cursor into its list: the next four words in registers (w0 is the current one), the four after them
// requested ahead.  The loads of the block behind are issued when a block becomes current, so they have four
// advances of this lane to land.
struct PmCursor {
    const uint64_t *w;
    uint32_t next, end;   // index of the first word of the block that is still to be requested; list length
    uint64_t a0, a1, a2, a3, b0, b1, b2, b3;
    uint32_t left;        // words of the current block not yet consumed (incl. a0)

    __device__ __forceinline__ void fetch_b()
    {
        const uint32_t p = next;
        b0 = p < end ? w[p] : PM_SENT;
        b1 = p + 1 < end ? w[p + 1] : PM_SENT;
        b2 = p + 2 < end ? w[p + 2] : PM_SENT;
        b3 = p + 3 < end ? w[p + 3] : PM_SENT;
        next = p + 4 < end ? p + 4 : end;
    }
    __device__ __forceinline__ void seek(const uint64_t *words, uint32_t n, uint32_t pos)
    {
        w = words; end = n; next = pos < n ? pos : n;
        fetch_b();
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        left = 4;
        fetch_b();
    }
    __device__ __forceinline__ void advance()
    {
        a0 = a1; a1 = a2; a2 = a3; a3 = PM_SENT;
        if (--left == 0) {
            a0 = b0; a1 = b1; a2 = b2; a3 = b3;
            left = 4;
            fetch_b();
        }
    }
};

__device__ __forceinline__ uint32_t pm_lower_bound(const uint64_t *w, uint32_t n, uint64_t key)
{
    uint32_t a = 0, b = n;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (w[mid] < key) a = mid + 1; else b = mid;
    }
    return a;
}

// pass 1: the word values that occur, per tile in LDS, then into the global occupancy bitmap
__global__ __launch_bounds__(PM_GROUP) void pm_mark_kernel(const PmList *__restrict__ lists, int n_samples,
                                                           const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                                           uint32_t tiles_per_range, uint64_t base,
                                                           unsigned long long *__restrict__ gbm, int single_group)
{
    __shared__ unsigned long long bm[PM_BMW];
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.y * PM_GROUP + threadIdx.x;
    const uint32_t t0 = blockIdx.x * tiles_per_range;
    const uint32_t t1 = t0 + tiles_per_range < n_tiles ? t0 + tiles_per_range : n_tiles;
    PmCursor cur;
    if (s < n_samples) {
        const PmList L = lists[s];
        cur.seek(L.words, (uint32_t)L.n, pm_lower_bound(L.words, (uint32_t)L.n, bounds[t0]));
    } else {
        cur.seek(nullptr, 0, 0);
    }
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        const uint32_t nbw = (uint32_t)((hi - lo + 63) >> 6);
        for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) bm[i] = 0;
        __syncthreads();
        for (;;) {
            const uint64_t cand = cur.a0 < hi ? cur.a0 : PM_SENT;
            const uint64_t m = pm_wave_min(cand);
            if (m == PM_SENT) break;
            if (lane == 0) atomicOr(&bm[(m - lo) >> 6], 1ull << ((m - lo) & 63));
            if (cand == m) cur.advance();
        }
        __syncthreads();
        unsigned long long *g = gbm + ((lo - base) >> 6);
        for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) {
            const unsigned long long v = bm[i];
            if (single_group) g[i] = v;
            else if (v) atomicOr(&g[i], v);
        }
        __syncthreads();
    }
}

__global__ void pm_popcount_kernel(const unsigned long long *__restrict__ gbm, uint64_t n_words, uint32_t *__restrict__ cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) cnt[i] = (uint32_t)__popcll(gbm[i]);
    else if (i == n_words) cnt[i] = 0;   // the scan's extra element: rank[n_words] = M
}

// rows of every tile, from the ranks at its two ends
__global__ void pm_tile_rows_kernel(const uint32_t *__restrict__ rank, const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                    uint64_t base, uint32_t *__restrict__ rows)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t a = (bounds[t] - base) >> 6, b = (bounds[t + 1] - base + 63) >> 6;
    rows[t] = rank[b] - rank[a];
}

// pass 2: the same stream; every wave stores its ballots into its column of the tile's block
__global__ __launch_bounds__(PM_GROUP) void pm_fill_kernel(const PmList *__restrict__ lists, int n_samples, int wpr,
                                                           const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                                           uint32_t tiles_per_range, uint64_t base,
                                                           const unsigned long long *__restrict__ gbm,
                                                           const uint32_t *__restrict__ rank, uint32_t r_cap,
                                                           uint32_t bmw_max, uint64_t *__restrict__ union_words,
                                                           uint64_t *__restrict__ bits)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long pm_lds[];
    unsigned long long *bm = pm_lds;                                        // bmw_max
    uint32_t *rk = reinterpret_cast<uint32_t *>(pm_lds + bmw_max);          // bmw_max + 1 (padded to even)
    unsigned long long *blk = pm_lds + bmw_max + ((bmw_max + 2) >> 1);      // cols x rows of the batch, column-major
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = blockIdx.y;
    const int s = group * PM_GROUP + threadIdx.x;
    const int cols = wpr - group * PM_COLS < PM_COLS ? wpr - group * PM_COLS : PM_COLS;   // u64 words of a row this group writes
    const uint32_t t0 = blockIdx.x * tiles_per_range;
    const uint32_t t1 = t0 + tiles_per_range < n_tiles ? t0 + tiles_per_range : n_tiles;
    const uint64_t *lw = nullptr;
    uint32_t ln = 0;
    if (s < n_samples) { const PmList L = lists[s]; lw = L.words; ln = (uint32_t)L.n; }
    PmCursor cur;
    uint32_t pos = lw ? pm_lower_bound(lw, ln, bounds[t0]) : 0;   // index of the lane's first word of the current tile
    cur.seek(lw, ln, pos);
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        const uint32_t nbw = (uint32_t)((hi - lo + 63) >> 6);
        const uint64_t w0 = (lo - base) >> 6;
        for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) bm[i] = gbm[w0 + i];
        for (uint32_t i = threadIdx.x; i <= nbw; i += blockDim.x) rk[i] = rank[w0 + i];
        __syncthreads();
        const uint32_t row0 = rk[0], rows = rk[nbw] - row0;
        uint32_t consumed = 0;   // words of this lane that belong to the tile (for the cursor of the next tile)
        for (uint32_t b0 = 0; b0 < rows; b0 += r_cap) {
            const uint32_t rb = rows - b0 < r_cap ? rows - b0 : r_cap;
            for (uint32_t i = threadIdx.x; i < rb * (uint32_t)cols; i += blockDim.x) blk[i] = 0;
            __syncthreads();
            if (b0 > 0) cur.seek(lw, ln, pos);   // a tile with more rows than the block holds is streamed once per batch
            uint32_t adv = 0;
            for (;;) {
                const uint64_t cand = cur.a0 < hi ? cur.a0 : PM_SENT;
                const uint64_t m = pm_wave_min(cand);
                if (m == PM_SENT) break;
                const bool hit = cand == m;
                const uint64_t mask = __ballot(hit);
                const uint32_t i = (uint32_t)(m - lo);
                const uint32_t r = rk[i >> 6] - row0 + (uint32_t)__popcll(bm[i >> 6] & ((1ull << (i & 63)) - 1ull)) - b0;
                if (lane == 0 && r < rb && wave < cols) blk[(uint32_t)wave * rb + r] = mask;   // r is unsigned: rows of earlier batches wrap
                if (hit) { cur.advance(); adv++; }
            }
            consumed = adv;
            __syncthreads();
            uint64_t *dst = bits + (uint64_t)(row0 + b0) * wpr + (uint64_t)group * PM_COLS;
            for (uint32_t e = threadIdx.x; e < rb * (uint32_t)cols; e += blockDim.x) {
                const uint32_t r = e / (uint32_t)cols, c = e % (uint32_t)cols;
                dst[(uint64_t)r * wpr + c] = blk[c * rb + r];
            }
            __syncthreads();
        }
        if (group == 0) {   // the union words of the tile: the set bits of its bitmap, in order
            for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) {
                unsigned long long v = bm[i];
                uint32_t r = rk[i];
                while (v) {
                    union_words[r++] = lo + ((uint64_t)i << 6) + (uint64_t)__builtin_ctzll(v);
                    v &= v - 1;
                }
            }
        }
        if (rows == 0) {   // nothing was streamed (no word of the tile in any sample of any group): nothing consumed
            consumed = 0;
        }
        pos += consumed;
        __syncthreads();   // bm / rk are rewritten by the next tile
    }
}

// evenly spaced entries of a few lists: the pilot the tile bounds are cut from
__global__ void pm_pilot_kernel(const PmList *__restrict__ lists, const int32_t *__restrict__ pick, int n_pick, uint32_t per_list,
                                uint64_t *__restrict__ out)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pick * per_list) return;
    const int li = (int)(g / per_list);
    const uint32_t j = (uint32_t)(g % per_list);
    const PmList L = lists[pick[li]];
    uint64_t v = PM_SENT;
    if (L.n) {
        uint64_t idx = (uint64_t)(((double)j + 0.5) * ((double)L.n / (double)per_list));
        if (idx >= L.n) idx = L.n - 1;
        v = L.words[idx];
    }
    out[g] = v;
}

}  // namespace

// Returns PSK_OK and sets *done = 1 when the merge build ran; *done = 0: not eligible, the caller takes the sort route.
int build_presence_merge(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done)
{
    *done = 0;
    if (getenv("PSK_NO_MERGE_PRESENCE")) return PSK_OK;
    const bool trace = getenv("PSK_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {   // PSK_TRACE: host-side phase times (each mark waits for the stream)
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[psk]   merge %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const int k = ctx->k, n = ctx->n_samples, wpr = ctx->wpr;
    if (2 * k > 40) return PSK_OK;
    const uint64_t space = 1ull << (2 * k);
    const uint64_t lo = ctx->slab_lo, hi = ctx->slab_hi ? ctx->slab_hi : space;
    const uint64_t base = lo & ~63ull, top = (hi + 63) & ~63ull;
    const uint64_t span = top - base;
    // one bit (+ half a rank byte) per word value of the slab: up to 2^34 values = 2 GB + 1 GB; rows are ranked in u32
    if (span > (1ull << 34)) return PSK_OK;
    if (total_pairs >= (1ull << 32) && span / 2 + (1ull << k) >= (1ull << 32)) return PSK_OK;
    for (int i = 0; i < n; i++)
        if (ctx->lists[i].n_unique >= (1ull << 32)) return PSK_OK;
    // ---- tile bounds: pair quantiles of a pilot, no tile wider than PM_BMW bitmap words ------------------------------
    uint64_t pairs_per_tile = 8192;
    if (const char *e = getenv("PSK_MERGE_TILE_PAIRS")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 64) pairs_per_tile = v; }
    uint64_t want = (total_pairs + pairs_per_tile - 1) / pairs_per_tile;
    if (want < 1) want = 1;
    if (want > (1ull << 24)) want = 1ull << 24;
    const int n_pick = n < 64 ? n : 64;
    std::vector<int32_t> pick(n_pick);
    for (int i = 0; i < n_pick; i++) pick[i] = (int32_t)(((int64_t)i * n) / n_pick);
    uint64_t per_list = (4 * want + n_pick - 1) / n_pick;
    if (per_list < 64) per_list = 64;
    if (per_list > (1u << 22)) per_list = 1u << 22;
    const uint64_t n_pilot = per_list * (uint64_t)n_pick;
    std::vector<PmList> refs(n);
    for (int i = 0; i < n; i++) { refs[i].words = ctx->lists[i].words; refs[i].n = ctx->lists[i].n_unique; }
    PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)n * sizeof(PmList) + (size_t)n_pick * 4 + 64));
    PmList *d_refs = ctx->starts.as<PmList>();
    int32_t *d_pick = reinterpret_cast<int32_t *>(d_refs + n);
    PSK_HIP(ctx, hipMemcpyAsync(d_refs, refs.data(), (size_t)n * sizeof(PmList), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d_pick, pick.data(), (size_t)n_pick * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->keysA, n_pilot * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, n_pilot * 8));
    pm_pilot_kernel<<<div_up(n_pilot, 256), 256, 0, ctx->stream>>>(d_refs, d_pick, n_pick, (uint32_t)per_list, ctx->keysA.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    uint64_t *sorted = nullptr;
    PSK_TRY(dev_radix_sort_u64(ctx, ctx->keysA.as<uint64_t>(), ctx->keysB.as<uint64_t>(), n_pilot, 0, 64, &sorted));   // all 64 bits: empty lists contribute sentinels, which must sort last
    std::vector<uint64_t> pilot(n_pilot);
    PSK_HIP(ctx, hipMemcpyAsync(pilot.data(), sorted, n_pilot * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // also: refs / pick (host) have been copied
    mark("pilot + sort");
    while (!pilot.empty() && pilot.back() == PM_SENT) pilot.pop_back();   // empty lists contributed sentinels
    std::vector<uint64_t> bounds;
    bounds.push_back(base);
    const uint64_t max_gap = (uint64_t)PM_BMW * 64;
    auto push_bound = [&](uint64_t b) {   // b: multiple of 64 in (bounds.back(), top]; gaps wider than a tile may be are cut evenly
        uint64_t prev = bounds.back();
        if (b <= prev) return;
        const uint64_t gap = b - prev;
        if (gap > max_gap) {
            const uint64_t parts = (gap + max_gap - 1) / max_gap;
            for (uint64_t q = 1; q < parts; q++) {
                const uint64_t mid = (prev + (gap * q) / parts) & ~63ull;
                if (mid > bounds.back() && mid < b) bounds.push_back(mid);
            }
        }
        bounds.push_back(b);
    };
    if (!pilot.empty())
        for (uint64_t j = 1; j < want; j++) {
            const uint64_t v = pilot[(size_t)((pilot.size() * j) / want)] & ~63ull;
            if (v > base && v < top) push_bound(v);
        }
    push_bound(top);
    const uint64_t n_tiles64 = bounds.size() - 1;
    if (n_tiles64 > (1ull << 26)) return PSK_OK;   // a sparse word space cut into tiles of 2^17 values: not this route's case
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    uint32_t bmw_max = 1;
    for (uint32_t t = 0; t < n_tiles; t++) bmw_max = std::max<uint32_t>(bmw_max, (uint32_t)((bounds[t + 1] - bounds[t] + 63) >> 6));
    // ---- launch shape ---------------------------------------------------------------------------------------------
    const int n_groups = (n + PM_GROUP - 1) / PM_GROUP;
    const int threads = n >= PM_GROUP ? PM_GROUP : ((n + 63) / 64) * 64;
    uint64_t n_ranges = ((uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8 * (PM_GROUP / threads)) / n_groups;
    if (const char *e = getenv("PSK_MERGE_RANGES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1) n_ranges = v; }
    if (n_ranges > n_tiles) n_ranges = n_tiles;
    if (n_ranges < 1) n_ranges = 1;
    const uint32_t tiles_per_range = (uint32_t)((n_tiles + n_ranges - 1) / n_ranges);
    n_ranges = (n_tiles + tiles_per_range - 1) / tiles_per_range;
    // ---- buffers: bounds | occupancy bitmap | ranks | rows per tile ---------------------------------------------------
    const uint64_t n_bmw = span >> 6;
    PSK_TRY(dev_reserve(ctx, ctx->flags, (size_t)(n_tiles + 1) * 8 + (size_t)n_tiles * 4 + 64));
    uint64_t *d_bounds = ctx->flags.as<uint64_t>();
    uint32_t *d_rows = reinterpret_cast<uint32_t *>(d_bounds + n_tiles + 1);
    PSK_TRY(dev_reserve(ctx, ctx->keysA, n_bmw * 8 + 64));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, (n_bmw + 1) * 4 + 64));
    unsigned long long *gbm = ctx->keysA.as<unsigned long long>();
    uint32_t *rank = ctx->keysB.as<uint32_t>();
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 2;
    PSK_HIP(ctx, hipMemcpyAsync(d_bounds, bounds.data(), (size_t)(n_tiles + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_groups > 1) PSK_HIP(ctx, hipMemsetAsync(gbm, 0, n_bmw * 8, ctx->stream));
    const dim3 grid((unsigned)n_ranges, (unsigned)n_groups);
    mark("bounds + buffers");
    pm_mark_kernel<<<grid, threads, 0, ctx->stream>>>(d_refs, n, d_bounds, n_tiles, tiles_per_range, base, gbm, n_groups == 1);
    PSK_HIP(ctx, hipGetLastError());
    mark("pm_mark");
    pm_popcount_kernel<<<div_up(n_bmw + 1, 256), 256, 0, ctx->stream>>>(gbm, n_bmw, rank);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(dev_exclusive_scan_u32(ctx, rank, rank, n_bmw + 1, d_m));
    pm_tile_rows_kernel<<<div_up(n_tiles, 256), 256, 0, ctx->stream>>>(rank, d_bounds, n_tiles, base, d_rows);
    PSK_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> rows(n_tiles);
    uint32_t m32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(rows.data(), d_rows, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(&m32, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `bounds` (host) has been copied as well
    mark("ranks");
    const uint64_t M = m32;
    uint32_t rows_max = 0;
    for (uint32_t t = 0; t < n_tiles; t++) rows_max = std::max(rows_max, rows[t]);
    PSK_TRY(dev_reserve(ctx, ctx->union_words, (M ? M : 1) * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, (M ? M : 1) * (uint64_t)wpr * 8));
    mark("alloc matrix");
    if (M) {
        const int cols0 = wpr < PM_COLS ? wpr : PM_COLS;
        const size_t head = (size_t)bmw_max * 8 + (size_t)((bmw_max + 2) >> 1) * 8;
        uint32_t r_cap = (uint32_t)((PM_LDS_MAX - head) / ((size_t)cols0 * 8));
        if (const char *e = getenv("PSK_MERGE_RCAP")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < r_cap) r_cap = v; }   // tests: force batches
        const uint32_t rb_max = rows_max < r_cap ? rows_max : r_cap;
        const size_t lds = head + (size_t)cols0 * rb_max * 8;
        if (lds > 64 * 1024)
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pm_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds));
        pm_fill_kernel<<<grid, threads, lds, ctx->stream>>>(d_refs, n, wpr, d_bounds, n_tiles, tiles_per_range, base, gbm, rank, r_cap,
                                                           bmw_max, ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark("pm_fill");
    if (getenv("PSK_TRACE"))
        fprintf(stderr, "[psk] merge build: %u tiles (%llu pairs each wanted), %llu ranges x %d groups of %d threads, widest tile %u bitmap words, most rows %u\n",
                n_tiles, (unsigned long long)pairs_per_tile, (unsigned long long)n_ranges, n_groups, threads, bmw_max, rows_max);
    *n_kmers = M;
    *done = 1;
    return PSK_OK;
}
