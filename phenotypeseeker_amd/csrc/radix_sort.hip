// Stable LSD radix sort of u64 keys, 8-bit digits.  Per pass:
//   radix_hist_kernel     per-tile digit histogram (LDS atomics)            -> hist[digit][tile]
//   dev_exclusive_scan    global digit/tile offsets
//   radix_scatter_kernel  wave-level multi-split ranking (8 ballots per key), keys bucketed by
//                         digit in LDS, then spilled to HBM in digit runs (coalesced stores)
// Used for the per-sample k-mer sort (a1) and the (word, sample) pair sort that yields the
// union and the presence matrix (a2+a3).  Bandwidth-bound integer work: no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_KPT = 16;                      // keys per thread
constexpr int RS_TILE = RS_THREADS * RS_KPT;    // 4096 keys per workgroup
constexpr int RS_RADIX = 256;

__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                                 int shift, uint32_t dmask, uint32_t n_tiles,
                                                                 uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_KPT; j++) {
        uint64_t i = base + (uint64_t)j * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// HAS_VAL: a u32 payload travels with every key (the (word, sample) pair sort when word and sample do
// not fit one u64, i.e. 2k + ceil(log2 N) > 64).
template <bool HAS_VAL>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const uint64_t *__restrict__ src,
                                                                    uint64_t *__restrict__ dst, uint64_t n, int shift,
                                                                    uint32_t dmask, uint32_t n_tiles,
                                                                    const uint32_t *__restrict__ hist_scanned,
                                                                    const uint32_t *__restrict__ vsrc,
                                                                    uint32_t *__restrict__ vdst)
{
    __shared__ uint64_t stage[RS_TILE];            // 32 KiB: keys bucketed by digit
    __shared__ uint32_t vstage[HAS_VAL ? RS_TILE : 1];
    __shared__ uint32_t wh[RS_WAVES][RS_RADIX];    // per-wave digit counters -> per-wave bucket bases
    __shared__ uint32_t gbase[RS_RADIX];           // global base of digit d minus its tile-local start
    __shared__ uint32_t scan_lds[RS_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t tile_base = (uint64_t)blockIdx.x * RS_TILE;
    const uint32_t n_valid = (uint32_t)((n - tile_base < (uint64_t)RS_TILE) ? (n - tile_base) : RS_TILE);

#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) wh[w][tid] = 0;
    __syncthreads();

    uint64_t key[RS_KPT];
    uint32_t rank[RS_KPT];
    uint32_t val[HAS_VAL ? RS_KPT : 1];
    const uint32_t wave_base = wid * (64 * RS_KPT);
#pragma unroll
    for (int r = 0; r < RS_KPT; r++) {
        const uint32_t li = wave_base + r * 64 + lane;  // position inside the tile, memory order
        const bool ok = li < n_valid;
        key[r] = ok ? src[tile_base + li] : ~0ull;
        if (HAS_VAL) val[r] = ok ? vsrc[tile_base + li] : 0u;
        const uint32_t d = ok ? ((uint32_t)(key[r] >> shift) & dmask) : (RS_RADIX - 1);
        // lanes holding the same digit (wave-level multi-split)
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t pre = __popcll(same & psk_lanemask_lt(lane));
        const uint32_t old = wh[wid][d];
        rank[r] = old + pre;
        if (pre == 0) wh[wid][d] = old + __popcll(same);
    }
    __syncthreads();

    // digit d = tid: totals over waves, tile-local start of each digit, per-wave bases
    {
        uint32_t c[RS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { c[w] = wh[w][tid]; tot += c[w]; }
        uint32_t all;
        const uint32_t dstart = psk_block_excl_scan_u32<RS_THREADS>(tot, &all, scan_lds);
        uint32_t acc = dstart;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { wh[w][tid] = acc; acc += c[w]; }
        gbase[tid] = hist_scanned[(uint64_t)tid * n_tiles + blockIdx.x] - dstart;
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < RS_KPT; r++) {
        const uint32_t li = wave_base + r * 64 + lane;
        const uint32_t d = (li < n_valid) ? ((uint32_t)(key[r] >> shift) & dmask) : (RS_RADIX - 1);
        stage[wh[wid][d] + rank[r]] = key[r];
        if (HAS_VAL) vstage[wh[wid][d] + rank[r]] = val[r];
    }
    __syncthreads();

    // spill: consecutive threads write consecutive addresses inside each digit run
#pragma unroll
    for (int j = 0; j < RS_KPT; j++) {
        const uint32_t i = j * RS_THREADS + tid;
        if (i < n_valid) {
            const uint64_t kx = stage[i];
            const uint32_t d = (uint32_t)(kx >> shift) & dmask;
            dst[(uint64_t)(gbase[d] + i)] = kx;
            if (HAS_VAL) vdst[(uint64_t)(gbase[d] + i)] = vstage[i];
        }
    }
}

}  // namespace

int dev_radix_sort_u64(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint64_t n, int bit_lo, int bit_hi,
                       uint64_t **sorted_out)
{
    return dev_radix_sort_kv(ctx, a, b, nullptr, nullptr, n, bit_lo, bit_hi, sorted_out, nullptr);
}

int dev_radix_sort_kv(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint32_t *va, uint32_t *vb, uint64_t n, int bit_lo,
                      int bit_hi, uint64_t **sorted_out, uint32_t **sorted_vals_out)
{
    *sorted_out = a;
    if (sorted_vals_out) *sorted_vals_out = va;
    if (n == 0 || bit_hi <= bit_lo) return PSK_OK;
    if (n >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "radix sort: %llu keys exceed the 2^32 limit",
                                            (unsigned long long)n);
    const uint32_t n_tiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const uint64_t hist_n = (uint64_t)RS_RADIX * n_tiles;
    PSK_TRY(dev_reserve(ctx, ctx->hist, hist_n * sizeof(uint32_t)));
    uint32_t *hist = ctx->hist.as<uint32_t>();
    uint64_t *src = a, *dst = b;
    uint32_t *vsrc = va, *vdst = vb;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        const int nb = (bit_hi - shift < 8) ? (bit_hi - shift) : 8;
        const uint32_t dmask = (1u << nb) - 1u;
        radix_hist_kernel<<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, n, shift, dmask, n_tiles, hist);
        PSK_HIP(ctx, hipGetLastError());
        PSK_TRY(dev_exclusive_scan_u32(ctx, hist, hist, hist_n, nullptr));
        if (va)
            radix_scatter_kernel<true><<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, dst, n, shift, dmask, n_tiles, hist,
                                                                               vsrc, vdst);
        else
            radix_scatter_kernel<false><<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, dst, n, shift, dmask, n_tiles,
                                                                                hist, nullptr, nullptr);
        PSK_HIP(ctx, hipGetLastError());
        uint64_t *t = src; src = dst; dst = t;
        uint32_t *vt = vsrc; vsrc = vdst; vdst = vt;
    }
    *sorted_out = src;
    if (sorted_vals_out) *sorted_vals_out = vsrc;
    return PSK_OK;
}
