// Wave64 / workgroup helpers shared by the kernels (gfx950: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ uint32_t psk_wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one u32 per thread over a workgroup of NT threads (NT multiple of 64, <= 1024).
// lds must hold NT/64 u32.  *total gets the workgroup sum in every thread.
template <int NT>
__device__ __forceinline__ uint32_t psk_block_excl_scan_u32(uint32_t v, uint32_t *total, uint32_t *lds)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = psk_wave_incl_scan_u32(v, lane);
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint32_t s = lds[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

__device__ __forceinline__ uint64_t psk_lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

__device__ __forceinline__ double psk_shfl_xor_f64(double v, int d)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, d, 64);
    hi = __shfl_xor(hi, d, 64);
    return __hiloint2double(hi, lo);
}
