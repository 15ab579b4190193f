// The k-base windows of a tile of the clean stream without a byte-by-byte roll: shared by the direct-address counting of
// small word spaces (dense_count.hip, k <= 13) and the bucketed sort of 32-bit words (bucket_count.hip, k = 14..16).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int KW_SEG = 32;   // window ends per thread

// ---- the 32 window ends of a thread, without a byte-by-byte roll ------------------------------------------------
// A thread owns 32 consecutive bytes of the clean stream and reads the 16 before them (K - 1 <= 15).  The 48 bytes
// become three bit streams, four bytes per multiply: F (2 bits per base, earlier bases more significant), R (the
// complemented codes, later bases more significant) and B (1 bit per byte: not a base, i.e. a window break).  The
// forward word of the window that ends at byte p is then one funnel shift of F, its reverse complement one funnel
// shift of R, and "no break inside" one funnel shift of B -- with K a template parameter every shift amount is a
// literal.  (Byte-by-byte rolling cost ~30 VALU instructions per window end: 13 us of the 17 us of the first cut's
// histogram kernel, measured by switching parts of the kernels off.)
struct Streams {
    uint32_t F[3], R[4], B[2];
};

__device__ __forceinline__ void pack4(uint32_t w, uint32_t &yf, uint32_t &yr, uint32_t &yb)
{
    const uint32_t x = ((w >> 1) ^ (w >> 2)) & 0x03030303u;   // A/a 0, C/c 1, G/g 2, T/t/U/u 3, one code per byte
    yf = (x * 0x40100401u) >> 24;                             // b0 << 6 | b1 << 4 | b2 << 2 | b3
    yr = ((x ^ 0x03030303u) * 0x01041040u) >> 24;             // ~b0 | ~b1 << 2 | ~b2 << 4 | ~b3 << 6
    yb = (((~w >> 6) & 0x01010101u) * 0x01020408u) >> 24;     // bit i: byte i has bit 6 clear (only '\n' in a clean stream)
}

__device__ __forceinline__ void load_streams(Streams &st, const uint8_t *__restrict__ clean, uint64_t len, uint64_t s)
{
    uint32_t raw[12];
#pragma unroll
    for (int j = 0; j < 12; j++) raw[j] = 0x0a0a0a0au;   // beyond either end of the buffer: breaks
    if (s < len) {
        const uint4 *p = reinterpret_cast<const uint4 *>(clean + s);
        const uint4 a = p[0], b = p[1];
        raw[4] = a.x; raw[5] = a.y; raw[6] = a.z; raw[7] = a.w;
        raw[8] = b.x; raw[9] = b.y; raw[10] = b.z; raw[11] = b.w;
        if (s >= 16) {
            const uint4 h = *reinterpret_cast<const uint4 *>(clean + s - 16);
            raw[0] = h.x; raw[1] = h.y; raw[2] = h.z; raw[3] = h.w;
        }
    }
    st.B[0] = st.B[1] = 0;
    st.R[3] = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {   // 16 bytes = one dword of F and of R
        uint32_t f = 0, r = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint32_t yf, yr, yb;
            pack4(raw[4 * q + e], yf, yr, yb);
            f |= yf << (8 * (3 - e));
            r |= yr << (8 * e);
            const int bit = 16 * q + 4 * e;   // byte position of the group's first byte
            st.B[bit >> 5] |= (yb & 0xfu) << (bit & 31);
        }
        st.F[2 - q] = f;   // position p sits at bits 2 * (47 - p) of F ...
        st.R[q] = r;       // ... and at bits 2 * p of R
    }
}

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, int sh)   // (hi:lo) >> sh, sh in 0..31, low dword
{
    return sh == 0 ? lo : __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
}

// canonical word of the window that ends at byte 16 + J of the thread's 48; false when a break lies inside it
template <int K, int J>
__device__ __forceinline__ bool window(const Streams &st, uint32_t &w)
{
    constexpr uint32_t mask = K >= 16 ? 0xffffffffu : (1u << (2 * (K >= 16 ? 0 : K))) - 1u;
    constexpr int sf = 2 * (31 - J);            // bit offset of the window in F
    constexpr int sr = 2 * (17 + J - K);        // ... in R
    constexpr int sb = 17 + J - K;              // ... in B
    const uint32_t fw = funnel(sf / 32 + 1 < 3 ? st.F[sf / 32 + 1] : 0u, st.F[sf / 32], sf & 31) & mask;
    const uint32_t rc = funnel(st.R[sr / 32 + 1], st.R[sr / 32], sr & 31) & mask;
    const uint32_t bm = (sb < 32 ? funnel(st.B[1], st.B[0], sb) : (st.B[1] >> (sb - 32))) & ((1u << K) - 1u);
    w = fw < rc ? fw : rc;
    return bm == 0;
}

template <int K, int J>
struct ForEachWindow {
    template <class F>
    static __device__ __forceinline__ void run(const Streams &st, uint32_t lo, uint32_t hi, F &&f)
    {
        uint32_t w;
        const bool ok = window<K, J>(st, w) && w >= lo && w < hi;
        f(J, ok, w);
        ForEachWindow<K, J + 1>::run(st, lo, hi, f);
    }
};
template <int K>
struct ForEachWindow<K, KW_SEG> {
    template <class F>
    static __device__ __forceinline__ void run(const Streams &, uint32_t, uint32_t, F &&) {}
};


}  // namespace
