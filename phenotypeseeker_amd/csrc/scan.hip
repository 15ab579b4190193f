// Device-wide exclusive prefix sum over u32 (reduce / recurse / apply), wave64 shuffles + LDS.
// Utility for the radix sort (digit offsets) and the run-length / row-index steps.
#include "psk_internal.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix,
// *block_total gets the sum (valid in all threads)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *block_total, uint32_t *lds /*>=8*/)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint32_t s = lds[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return woff + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                                    uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    if (base + SCAN_ITEMS <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; j++) {
            uint4 v = p[j];
            s += v.x + v.y + v.z + v.w;
        }
    } else {
        for (int j = 0; j < SCAN_ITEMS; j++)
            if (base + j < n) s += in[base + j];
    }
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// One block scans up to SCAN_TILE values in place (top of the recursion).
__global__ __launch_bounds__(SCAN_THREADS) void scan_single_kernel(const uint32_t *__restrict__ in,
                                                                    uint32_t *__restrict__ out, uint32_t n,
                                                                    uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t lds[8];
    uint32_t v[SCAN_ITEMS];
    const uint32_t base = threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, &tot, lds);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
    if (threadIdx.x == 0 && total_out) *total_out = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t *__restrict__ in,
                                                                   uint32_t *__restrict__ out, uint64_t n,
                                                                   const uint32_t *__restrict__ bsum_scanned)
{
    __shared__ uint32_t lds[8];
    uint32_t v[SCAN_ITEMS];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    const bool full = base + SCAN_ITEMS <= n;
    if (full) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; j++) {
            uint4 q = p[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) v[j] = (base + j < n) ? in[base + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += v[j];
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, &tot, lds) + bsum_scanned[blockIdx.x];
    if (full) {
        uint4 *q = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; j++) {
            uint4 o;
            o.x = ex; ex += v[4 * j];
            o.y = ex; ex += v[4 * j + 1];
            o.z = ex; ex += v[4 * j + 2];
            o.w = ex; ex += v[4 * j + 3];
            q[j] = o;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            if (base + j < n) out[base + j] = ex;
            ex += v[j];
        }
    }
}

// scratch layout inside ctx->scan_tmp: level sums one after another
int scan_rec(psk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_out, uint32_t *scratch)
{
    if (n <= (uint64_t)SCAN_TILE) {
        scan_single_kernel<<<1, SCAN_THREADS, 0, ctx->stream>>>(in, out, (uint32_t)n, total_out);
        PSK_HIP(ctx, hipGetLastError());
        return PSK_OK;
    }
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *bsum = scratch;
    scan_reduce_kernel<<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(in, n, bsum);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(scan_rec(ctx, bsum, bsum, nb, total_out, scratch + ((nb + 3) & ~3ull)));
    scan_apply_kernel<<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(in, out, n, bsum);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

}  // namespace

int dev_exclusive_scan_u32(psk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_out)
{
    if (n == 0) {
        if (total_out) PSK_HIP(ctx, hipMemsetAsync(total_out, 0, 4, ctx->stream));
        return PSK_OK;
    }
    // scratch: sum over levels of ceil(n / 4096^l), padded
    uint64_t need = 0, m = n;
    while (m > (uint64_t)SCAN_TILE) {
        m = (m + SCAN_TILE - 1) / SCAN_TILE;
        need += (m + 3) & ~3ull;
    }
    PSK_TRY(dev_reserve(ctx, ctx->scan_tmp, (need + 4) * sizeof(uint32_t)));
    return scan_rec(ctx, in, out, n, total_out, ctx->scan_tmp.as<uint32_t>());
}
