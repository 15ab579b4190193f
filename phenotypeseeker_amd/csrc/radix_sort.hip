// Stable LSD radix sort of u64 keys, 8- or 9-bit digits.  Per pass:
//   radix_hist_kernel        per-tile digit histogram (LDS atomics)                  -> hist[digit][tile]
//   radix_digit_scan_kernel  one workgroup per digit: exclusive scan over the tiles  -> hist in place,
//                            digit totals aside (the digit bases are a 256/512-entry scan that every
//                            scatter workgroup redoes in LDS -- cheaper than another launch)
//   radix_scatter_kernel     wave-level multi-split ranking (one ballot per digit bit), keys bucketed by
//                            digit in LDS, then spilled to HBM in digit runs (coalesced stores)
// The significant bits are spread evenly over ceil(bits / 9) passes, so 26 bits (k = 13) take 9+9+8 and
// 34 bits (k = 13 words + 8 sample bits) 9+9+8+8 instead of four / five 8-bit passes.
// Used for the per-sample k-mer sort (a1) and the (word, sample) pair sort that yields the
// union and the presence matrix (a2+a3).  Bandwidth-bound integer work: no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_KPT = 16;                      // keys per thread
constexpr int RS_TILE = RS_THREADS * RS_KPT;    // 4096 keys per workgroup

template <int RB>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n_host,
                                                                 const uint32_t *__restrict__ n_dev, int shift,
                                                                 uint32_t dmask, uint32_t n_tiles,
                                                                 uint32_t *__restrict__ hist)
{
    constexpr int RADIX = 1 << RB, DPT = RADIX / RS_THREADS;  // digits per thread
    const uint64_t n = n_dev ? (uint64_t)*n_dev : n_host;      // n_dev: key count produced on the device, <= n_host
    __shared__ uint32_t h[RADIX];
#pragma unroll
    for (int e = 0; e < DPT; e++) h[e * RS_THREADS + threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_KPT; j++) {
        uint64_t i = base + (uint64_t)j * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < DPT; e++) {
        const int d = e * RS_THREADS + threadIdx.x;
        hist[(uint64_t)d * n_tiles + blockIdx.x] = h[d];
    }
}

// workgroup d: exclusive scan of hist[d][0 .. n_tiles) in place, total[d] = the digit's key count
__global__ __launch_bounds__(RS_THREADS) void radix_digit_scan_kernel(uint32_t *__restrict__ hist, uint32_t n_tiles,
                                                                       uint32_t *__restrict__ total)
{
    __shared__ uint32_t scan_lds[RS_WAVES];
    uint32_t *row = hist + (uint64_t)blockIdx.x * n_tiles;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += RS_THREADS * 4) {
        // four consecutive tiles per thread
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t t = t0 + threadIdx.x * 4 + e;
            v[e] = t < n_tiles ? row[t] : 0u;
            sum += v[e];
        }
        uint32_t all;
        uint32_t ex = carry + psk_block_excl_scan_u32<RS_THREADS>(sum, &all, scan_lds);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t t = t0 + threadIdx.x * 4 + e;
            if (t < n_tiles) row[t] = ex;
            ex += v[e];
        }
        carry += all;
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

// HAS_VAL: a u32 payload travels with every key (the (word, sample) pair sort when word and sample do
// not fit one u64, i.e. 2k + ceil(log2 N) > 64).
template <int RB, bool HAS_VAL>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const uint64_t *__restrict__ src,
                                                                    uint64_t *__restrict__ dst, uint64_t n_host,
                                                                    const uint32_t *__restrict__ n_dev, int shift,
                                                                    uint32_t dmask, uint32_t n_tiles,
                                                                    const uint32_t *__restrict__ hist_scanned,
                                                                    const uint32_t *__restrict__ total,
                                                                    const uint32_t *__restrict__ vsrc,
                                                                    uint32_t *__restrict__ vdst)
{
    constexpr int RADIX = 1 << RB, DPT = RADIX / RS_THREADS;
    __shared__ uint64_t stage[RS_TILE];            // 32 KiB: keys bucketed by digit
    __shared__ uint32_t vstage[HAS_VAL ? RS_TILE : 1];
    __shared__ uint32_t wh[RS_WAVES][RADIX];       // per-wave digit counters -> per-wave bucket bases
    __shared__ uint32_t gbase[RADIX];              // global base of digit d minus its tile-local start
    __shared__ uint32_t scan_lds[RS_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t n = n_dev ? (uint64_t)*n_dev : n_host;
    const uint64_t tile_base = (uint64_t)blockIdx.x * RS_TILE;
    if (tile_base >= n) return;  // tiles beyond a device-side count (the grid covers the host's upper bound)
    const uint32_t n_valid = (uint32_t)((n - tile_base < (uint64_t)RS_TILE) ? (n - tile_base) : RS_TILE);

#pragma unroll
    for (int w = 0; w < RS_WAVES; w++)
#pragma unroll
        for (int e = 0; e < DPT; e++) wh[w][e * RS_THREADS + tid] = 0;
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
        const uint32_t d = ok ? ((uint32_t)(key[r] >> shift) & dmask) : (RADIX - 1);
        // lanes holding the same digit (wave-level multi-split)
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < RB; b++) {
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

    // thread tid owns digits tid*DPT .. tid*DPT+DPT-1 (consecutive, so one block scan orders all of them):
    // totals over waves, tile-local start of each digit, per-wave bases, and the global digit bases
    {
        uint32_t c[DPT][RS_WAVES], tot[DPT], gt[DPT], tsum = 0, gsum = 0;
#pragma unroll
        for (int e = 0; e < DPT; e++) {
            const int d = tid * DPT + e;
            tot[e] = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { c[e][w] = wh[w][d]; tot[e] += c[e][w]; }
            gt[e] = total[d];
            tsum += tot[e];
            gsum += gt[e];
        }
        uint32_t all;
        uint32_t dstart = psk_block_excl_scan_u32<RS_THREADS>(tsum, &all, scan_lds);
        uint32_t dbase = psk_block_excl_scan_u32<RS_THREADS>(gsum, &all, scan_lds);
#pragma unroll
        for (int e = 0; e < DPT; e++) {
            const int d = tid * DPT + e;
            uint32_t acc = dstart;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { wh[w][d] = acc; acc += c[e][w]; }
            gbase[d] = dbase + hist_scanned[(uint64_t)d * n_tiles + blockIdx.x] - dstart;
            dstart += tot[e];
            dbase += gt[e];
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < RS_KPT; r++) {
        const uint32_t li = wave_base + r * 64 + lane;
        const uint32_t d = (li < n_valid) ? ((uint32_t)(key[r] >> shift) & dmask) : (RADIX - 1);
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

template <int RB>
int radix_pass(psk_ctx *ctx, const uint64_t *src, uint64_t *dst, const uint32_t *vsrc, uint32_t *vdst, uint64_t n,
               const uint32_t *n_dev, int shift, int nb, uint32_t n_tiles, uint32_t *hist, uint32_t *total)
{
    const uint32_t dmask = (1u << nb) - 1u;
    radix_hist_kernel<RB><<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, n, n_dev, shift, dmask, n_tiles, hist);
    PSK_HIP(ctx, hipGetLastError());
    radix_digit_scan_kernel<<<1 << RB, RS_THREADS, 0, ctx->stream>>>(hist, n_tiles, total);
    PSK_HIP(ctx, hipGetLastError());
    if (vsrc)
        radix_scatter_kernel<RB, true><<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, dst, n, n_dev, shift, dmask, n_tiles, hist, total,
                                                                              vsrc, vdst);
    else
        radix_scatter_kernel<RB, false><<<n_tiles, RS_THREADS, 0, ctx->stream>>>(src, dst, n, n_dev, shift, dmask, n_tiles, hist,
                                                                               total, nullptr, nullptr);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

}  // namespace

int dev_radix_sort_u64(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint64_t n, int bit_lo, int bit_hi,
                       uint64_t **sorted_out, const uint32_t *n_dev)
{
    return dev_radix_sort_kv(ctx, a, b, nullptr, nullptr, n, bit_lo, bit_hi, sorted_out, nullptr, n_dev);
}

// n_dev != nullptr: the actual key count sits in device memory (<= n, which then only sizes the launches)
int dev_radix_sort_kv(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint32_t *va, uint32_t *vb, uint64_t n, int bit_lo,
                      int bit_hi, uint64_t **sorted_out, uint32_t **sorted_vals_out, const uint32_t *n_dev)
{
    *sorted_out = a;
    if (sorted_vals_out) *sorted_vals_out = va;
    if (n == 0 || bit_hi <= bit_lo) return PSK_OK;
    if (n >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "radix sort: %llu keys exceed the 2^32 limit",
                                            (unsigned long long)n);
    const uint32_t n_tiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const uint64_t hist_n = (uint64_t)512 * n_tiles + 512;  // [digit][tile] + the digit totals
    PSK_TRY(dev_reserve(ctx, ctx->hist, hist_n * sizeof(uint32_t)));
    uint32_t *hist = ctx->hist.as<uint32_t>();
    uint32_t *total = hist + (uint64_t)512 * n_tiles;
    uint64_t *src = a, *dst = b;
    uint32_t *vsrc = va, *vdst = vb;
    const int bits = bit_hi - bit_lo;
    const int n_pass = (bits + 8) / 9;
    const int base = bits / n_pass, rem = bits % n_pass;  // the first `rem` passes take one bit more
    int shift = bit_lo;
    for (int ps = 0; ps < n_pass; ps++) {
        const int nb = base + (ps < rem ? 1 : 0);
        if (nb > 8) PSK_TRY(radix_pass<9>(ctx, src, dst, vsrc, vdst, n, n_dev, shift, nb, n_tiles, hist, total));
        else PSK_TRY(radix_pass<8>(ctx, src, dst, vsrc, vdst, n, n_dev, shift, nb, n_tiles, hist, total));
        shift += nb;
        uint64_t *t = src; src = dst; dst = t;
        uint32_t *vt = vsrc; vsrc = vdst; vdst = vt;
    }
    *sorted_out = src;
    if (sorted_vals_out) *sorted_vals_out = vsrc;
    return PSK_OK;
}
