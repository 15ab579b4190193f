// Sort-free build of the union + presence matrix for small word spaces (a2 + a3, get_feature_vector / map_samples,
// modeling.py:317-380).  The per-sample lists are already sorted, so the word space of the run is cut into tiles of
// R consecutive word values, R chosen so that a tile's R x wpr bit block fits in LDS:
//   tile_starts   lower bound of every tile boundary in every sample's list (binary searches)
//   tile_count    per tile: which of its R words occur in any sample (R-bit LDS bitmap)  -> rows of the tile
//   (scan)        exclusive scan of the tile row counts -> first output row of every tile, M = total
//   tile_fill     per tile: every sample's segment sets its bits in the LDS block (ds_or), non-empty rows are
//                 ranked and written out -- union words ascending, rows packed
// Traffic: the lists are read twice (2 x 8 B per (word, sample) pair) and the matrix is written once, against
// ~96 B per pair for the pack + 4-pass radix sort + head-flag route (which stays for large k, where the tile
// table would not fit).  256 x 5 Mbp, k = 13: see DESIGN.md.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int PT_THREADS = 1024;  // one 128 KB tile per CU, 16 waves on it
constexpr size_t PT_LDS_MAX = 156 * 1024;  // of the CU's 160 KB

struct ListRef {
    const uint64_t *words;
    uint64_t n;
};

// start[s][t] = first index of list s whose word is >= lo + t * R   (t = 0 .. n_tiles inclusive)
__global__ void tile_starts_kernel(const ListRef *__restrict__ lists, int n_samples, uint64_t lo, uint64_t R,
                                   uint32_t n_tiles, uint32_t *__restrict__ start)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per = (uint64_t)n_tiles + 1;
    if (g >= per * (uint64_t)n_samples) return;
    const int s = (int)(g / per);
    const uint64_t t = g % per;
    const uint64_t *w = lists[s].words;
    const uint64_t n = lists[s].n;
    uint64_t a = 0, b = n;
    if (t == per - 1) {
        a = n;  // everything below the end of the run's range
    } else {
        const uint64_t key = lo + t * R;
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (w[mid] < key) a = mid + 1; else b = mid;
        }
    }
    start[(uint64_t)s * per + t] = (uint32_t)a;
}

// rows of every tile = number of distinct words of the tile over all samples
__global__ __launch_bounds__(PT_THREADS) void tile_count_kernel(const ListRef *__restrict__ lists, int n_samples,
                                                                uint64_t lo, uint32_t R, uint32_t n_tiles,
                                                                const uint32_t *__restrict__ start,
                                                                uint32_t *__restrict__ tile_rows, int lane_per_sample)
{
    extern __shared__ uint64_t occ[];  // R bits
    __shared__ uint32_t red[PT_THREADS / 64];
    const uint32_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t tile_lo = lo + (uint64_t)tile * R;
    const uint64_t per = (uint64_t)n_tiles + 1;
    for (uint32_t i = threadIdx.x; i < R / 64; i += PT_THREADS) occ[i] = 0;
    __syncthreads();
    if (lane_per_sample) {  // short segments (many samples, small genomes): one sample per thread, its segment in sequence
        for (int s = threadIdx.x; s < n_samples; s += PT_THREADS) {
            const uint32_t a = start[(uint64_t)s * per + tile], b = start[(uint64_t)s * per + tile + 1];
            const uint64_t *w = lists[s].words;
            for (uint32_t j = a; j < b; j++) {
                const uint32_t r = (uint32_t)(w[j] - tile_lo);
                atomicOr(reinterpret_cast<unsigned long long *>(&occ[r >> 6]), 1ull << (r & 63));
            }
        }
    } else {
        for (int s = wid; s < n_samples; s += PT_THREADS / 64) {
            const uint32_t a = start[(uint64_t)s * per + tile], b = start[(uint64_t)s * per + tile + 1];
            const uint64_t *w = lists[s].words;
            for (uint32_t j = a + lane; j < b; j += 64) {
                const uint32_t r = (uint32_t)(w[j] - tile_lo);
                atomicOr(reinterpret_cast<unsigned long long *>(&occ[r >> 6]), 1ull << (r & 63));
            }
        }
    }
    __syncthreads();
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < R / 64; i += PT_THREADS) c += __popcll(occ[i]);
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane == 0) red[wid] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < PT_THREADS / 64; w++) t += red[w];
        tile_rows[tile] = t;
    }
}

__global__ __launch_bounds__(PT_THREADS) void tile_fill_kernel(const ListRef *__restrict__ lists, int n_samples, int wpr,
                                                               uint64_t lo, uint32_t R, uint32_t n_tiles,
                                                               const uint32_t *__restrict__ start,
                                                               const uint32_t *__restrict__ tile_off,
                                                               uint64_t *__restrict__ union_words,
                                                               uint64_t *__restrict__ bits, int lane_per_sample)
{
    extern __shared__ uint64_t blk[];  // wpr x R words, column-major (blk[col][row])
    __shared__ uint32_t scan_lds[PT_THREADS / 64];
    const uint32_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t tile_lo = lo + (uint64_t)tile * R;
    const uint64_t per = (uint64_t)n_tiles + 1;
    const uint32_t out0 = tile_off[tile];
    const uint32_t n_rows = tile_off[tile + 1] - out0;
    if (n_rows == 0) return;  // uniform for the workgroup
    for (uint32_t i = threadIdx.x; i < R * (uint32_t)wpr; i += PT_THREADS) blk[i] = 0;
    __syncthreads();
    if (lane_per_sample) {
        for (int s = threadIdx.x; s < n_samples; s += PT_THREADS) {
            const uint32_t a = start[(uint64_t)s * per + tile], b = start[(uint64_t)s * per + tile + 1];
            const uint64_t *w = lists[s].words;
            const uint64_t bit = 1ull << (s & 63);
            const uint32_t col = (uint32_t)s >> 6;
            for (uint32_t j = a; j < b; j++) {
                const uint32_t r = (uint32_t)(w[j] - tile_lo);
                atomicOr(reinterpret_cast<unsigned long long *>(&blk[col * R + r]), bit);
            }
        }
    } else {
        for (int s = wid; s < n_samples; s += PT_THREADS / 64) {
            const uint32_t a = start[(uint64_t)s * per + tile], b = start[(uint64_t)s * per + tile + 1];
            const uint64_t *w = lists[s].words;
            const uint64_t bit = 1ull << (s & 63);
            const uint32_t col = (uint32_t)s >> 6;
            for (uint32_t j = a + lane; j < b; j += 64) {
                const uint32_t r = (uint32_t)(w[j] - tile_lo);
                atomicOr(reinterpret_cast<unsigned long long *>(&blk[col * R + r]), bit);  // column-major: random rows spread over the banks
            }
        }
    }
    __syncthreads();
    // rank the non-empty rows: thread t owns rows t * RPT .. (consecutive, so one block scan orders them), and lists
    // them in LDS; the rows are then copied out by the whole workgroup with consecutive threads on consecutive
    // addresses (a thread writing its own 32-byte rows was 13 of the build's 17 ms)
    uint32_t *row_of = reinterpret_cast<uint32_t *>(blk + (size_t)R * wpr);  // R entries behind the block
    // R is a power of two (host): a multiple of PT_THREADS, or -- many samples, wide rows -- a fraction of it, and
    // then only the first R threads own a row
    const uint32_t rpt = R >= PT_THREADS ? R / PT_THREADS : 1;
    uint32_t mine = 0;
    uint32_t nonempty = 0;  // bit q: row t * rpt + q is occupied (rpt <= 32)
    for (uint32_t q = 0; q < rpt; q++) {
        const uint32_t r = threadIdx.x * rpt + q;
        if (r >= R) break;
        uint64_t any = 0;
        for (int c = 0; c < wpr; c++) any |= blk[(uint32_t)c * R + r];
        if (any) { nonempty |= 1u << q; mine++; }
    }
    uint32_t all;
    uint32_t rank = psk_block_excl_scan_u32<PT_THREADS>(mine, &all, scan_lds);
    for (uint32_t q = 0; q < rpt; q++)
        if ((nonempty >> q) & 1) row_of[rank++] = threadIdx.x * rpt + q;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_rows; i += PT_THREADS) union_words[(uint64_t)out0 + i] = tile_lo + row_of[i];
    const uint32_t n_out = n_rows * (uint32_t)wpr;
    uint64_t *dst = bits + (uint64_t)out0 * wpr;
    for (uint32_t e = threadIdx.x; e < n_out; e += PT_THREADS)
        dst[e] = blk[(e % (uint32_t)wpr) * R + row_of[e / (uint32_t)wpr]];
}

}  // namespace

// Returns PSK_OK and sets *done = 1 when the tiled build ran; *done = 0 means "not eligible, use the sort route".
int build_presence_tiled(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done)
{
    *done = 0;
    if (getenv("PSK_NO_TILED_PRESENCE")) return PSK_OK;
    const int k = ctx->k, n = ctx->n_samples, wpr = ctx->wpr;
    if (2 * k > 40) return PSK_OK;
    const uint64_t space = 1ull << (2 * k);
    const uint64_t lo = ctx->slab_lo, hi = ctx->slab_hi ? ctx->slab_hi : space;
    const uint64_t span = hi - lo;
    // rows are numbered in u32: the union of this slab must stay below 2^32 rows, which the pair count or the
    // number of canonical words of the slab (at most span / 2 + 2^k palindromes) guarantees
    if (total_pairs >= (1ull << 32) && span / 2 + (1ull << k) >= (1ull << 32)) return PSK_OK;
    // tile size: the smallest LDS block (more workgroups per CU) whose tile table stays under 4 GB (and small next
    // to the lists, below); rows per thread are capped at 32 (a bit mask in tile_fill); with thousands of samples a
    // row is hundreds of bytes and a tile holds fewer rows than the workgroup has threads (down to 64)
    uint32_t R = 0;
    uint64_t n_tiles64 = 0;
    const size_t forced = getenv("PSK_TILE_LDS_KB") ? (size_t)atoi(getenv("PSK_TILE_LDS_KB")) * 1024 : 0;
    for (size_t lds_try = forced ? forced : PT_LDS_MAX; lds_try <= PT_LDS_MAX; lds_try *= 2) {
        uint32_t r = (uint32_t)(lds_try / ((size_t)wpr * 8 + 4));  // bit block + the 4-byte entry of the occupied-row list
        if (r < 64) { if (forced) break; continue; }  // not even 64 rows fit: too many samples for this block size
        uint32_t p2 = 64;
        while ((uint64_t)p2 * 2 <= r) p2 *= 2;
        r = p2;
        if (r > 32u * PT_THREADS) r = 32u * PT_THREADS;
        if ((size_t)r * ((size_t)wpr * 8 + 4) > lds_try) { if (forced) break; continue; }  // too many samples for this block size
        const uint64_t nt = (span + r - 1) / r;
        if (nt <= (1u << 22) && (nt + 1) * (uint64_t)n * 4 <= (4096ull << 20)) { R = r; n_tiles64 = nt; break; }
        if (forced) break;
    }
    if (R == 0) return PSK_OK;
    // ... and it must stay small next to the lists it indexes
    if ((n_tiles64 + 1) * (uint64_t)n > 4 * total_pairs + (1u << 20)) return PSK_OK;
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    std::vector<ListRef> refs(n);
    for (int i = 0; i < n; i++) {
        if (ctx->lists[i].n_unique >= (1ull << 32)) return PSK_OK;
        refs[i].words = ctx->lists[i].words;
        refs[i].n = ctx->lists[i].n_unique;
    }
    const uint64_t per = (uint64_t)n_tiles + 1;
    const size_t table_bytes = per * (size_t)n * 4;
    const size_t need = table_bytes + (size_t)n * sizeof(ListRef) + ((size_t)n_tiles + 2) * 4 + 256;
    PSK_TRY(dev_reserve(ctx, ctx->keysA, need));
    uint8_t *base = ctx->keysA.as<uint8_t>();
    uint32_t *start = reinterpret_cast<uint32_t *>(base);
    ListRef *d_refs = reinterpret_cast<ListRef *>(base + ((table_bytes + 15) & ~size_t(15)));
    uint32_t *tile_rows = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_refs) + (size_t)n * sizeof(ListRef));
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 2;
    PSK_HIP(ctx, hipMemcpyAsync(d_refs, refs.data(), (size_t)n * sizeof(ListRef), hipMemcpyHostToDevice, ctx->stream));
    tile_starts_kernel<<<div_up(per * (uint64_t)n, 256), 256, 0, ctx->stream>>>(d_refs, n, lo, R, n_tiles, start);
    PSK_HIP(ctx, hipGetLastError());
    // average segment (words of one sample inside one tile) below half a wave: one sample per thread instead
    const int lane_per_sample = total_pairs < (uint64_t)n * n_tiles * 32 ? 1 : 0;
    tile_count_kernel<<<n_tiles, PT_THREADS, R / 8, ctx->stream>>>(d_refs, n, lo, R, n_tiles, start, tile_rows, lane_per_sample);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemsetAsync(tile_rows + n_tiles, 0, 4, ctx->stream));  // the scan's extra element: tile_off[n_tiles] = M
    PSK_TRY(dev_exclusive_scan_u32(ctx, tile_rows, tile_rows, (uint64_t)n_tiles + 1, d_m));
    uint32_t m32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&m32, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `refs` (host) must outlive its copy as well
    const uint64_t M = m32;
    PSK_TRY(dev_reserve(ctx, ctx->union_words, M * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, M * (uint64_t)wpr * 8));
    if (M) {
        const size_t lds = (size_t)R * wpr * 8 + (size_t)R * 4;  // bit block + the list of occupied rows
        if (lds > 64 * 1024)
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(tile_fill_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        tile_fill_kernel<<<n_tiles, PT_THREADS, lds, ctx->stream>>>(d_refs, n, wpr, lo, R, n_tiles, start, tile_rows,
                                                                   ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>(), lane_per_sample);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_kmers = M;
    *done = 1;
    return PSK_OK;
}
