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

// ---- LDS-free cross-lane helpers (DPP + v_readlane) ---------------------------------------------
// ds_bpermute-based shuffles issue on the CU's single LDS pipe; a phase that is already LDS-heavy
// (t-test phase B) is better served by data-parallel-primitive moves and scalar lane reads.
__device__ __forceinline__ double psk_dpp_f64(double v, const int ctrl_tag)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_tag) {  // constant-folded: the builtin needs an immediate control word
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
    }
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double psk_readlane_f64(double v, int lane_uniform)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_uniform);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_uniform);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, result uniform.  After the two quad steps every lane of a quad holds the quad
// sum, so the mirrors act as xor-4 / xor-8 butterflies; the four 16-lane row sums are combined by
// scalar lane reads.  Requires all 64 lanes active.
__device__ __forceinline__ double psk_wave_sum_f64_dpp(double v)
{
    v += psk_dpp_f64(v, 0);
    v += psk_dpp_f64(v, 1);
    v += psk_dpp_f64(v, 2);
    v += psk_dpp_f64(v, 3);
    return (psk_readlane_f64(v, 0) + psk_readlane_f64(v, 16)) + (psk_readlane_f64(v, 32) + psk_readlane_f64(v, 48));
}

// Maximum over the 64 lanes, result uniform (same butterflies as the sum; NaN-free inputs).
__device__ __forceinline__ double psk_wave_max_f64_dpp(double v)
{
    v = fmax(v, psk_dpp_f64(v, 0));
    v = fmax(v, psk_dpp_f64(v, 1));
    v = fmax(v, psk_dpp_f64(v, 2));
    v = fmax(v, psk_dpp_f64(v, 3));
    return fmax(fmax(psk_readlane_f64(v, 0), psk_readlane_f64(v, 16)), fmax(psk_readlane_f64(v, 32), psk_readlane_f64(v, 48)));
}

__device__ __forceinline__ uint64_t psk_readlane_u64(uint64_t v, int lane_uniform)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_uniform);
    return ((uint64_t)hi << 32) | lo;
}
