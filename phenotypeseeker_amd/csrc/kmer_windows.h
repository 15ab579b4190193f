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

// ---- the same for word spaces beyond 32 bits (K = 17..32: 64-bit words; bucket_count.hip, r05) --------------------------------
// A thread still owns 32 consecutive bytes (window ends) and reads the 32 before them (K - 1 <= 31): 64 bytes, F and R of
// 128 bits each (position p of the 64 at bits 2 (63 - p) of F and at bits 2 p of R), B of 64 bits.  A word is two funnel shifts.
struct StreamsW {
    uint32_t F[4], R[5], B[3];
};

__device__ __forceinline__ void load_streams(StreamsW &st, const uint8_t *__restrict__ clean, uint64_t len, uint64_t s)
{
    uint32_t raw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) raw[j] = 0x0a0a0a0au;   // beyond either end of the buffer: breaks
    if (s < len) {
        const uint4 *p = reinterpret_cast<const uint4 *>(clean + s);
        const uint4 a = p[0], b = p[1];
        raw[8] = a.x; raw[9] = a.y; raw[10] = a.z; raw[11] = a.w;
        raw[12] = b.x; raw[13] = b.y; raw[14] = b.z; raw[15] = b.w;
        if (s >= 32) {
            const uint4 *h = reinterpret_cast<const uint4 *>(clean + s - 32);
            const uint4 c = h[0], d = h[1];
            raw[0] = c.x; raw[1] = c.y; raw[2] = c.z; raw[3] = c.w;
            raw[4] = d.x; raw[5] = d.y; raw[6] = d.z; raw[7] = d.w;
        }
    }
    st.B[0] = st.B[1] = st.B[2] = 0;
    st.R[4] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {   // 16 bytes = one dword of F and of R
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
        st.F[3 - q] = f;
        st.R[q] = r;
    }
}

// canonical word of the window that ends at byte 32 + J of the thread's 64; false when a break lies inside it
template <int K, int J>
__device__ __forceinline__ bool window(const StreamsW &st, uint64_t &w)
{
    static_assert(K >= 17 && K <= 32, "64-bit words: K = 17..32");
    constexpr uint64_t mask = K >= 32 ? ~0ull : (1ull << (2 * (K >= 32 ? 0 : K))) - 1ull;
    constexpr int sf = 2 * (31 - J);            // bit offset of the window in F (0..62)
    constexpr int sr = 2 * (33 + J - K);        // ... in R (2..94)
    constexpr int sb = 33 + J - K;              // ... in B (1..47)
    const uint32_t f_lo = funnel(st.F[sf / 32 + 1], st.F[sf / 32], sf & 31);
    const uint32_t f_hi = funnel(st.F[sf / 32 + 2], st.F[sf / 32 + 1], sf & 31);
    const uint32_t r_lo = funnel(st.R[sr / 32 + 1], st.R[sr / 32], sr & 31);
    const uint32_t r_hi = funnel(st.R[sr / 32 + 2], st.R[sr / 32 + 1], sr & 31);
    const uint64_t fw = (((uint64_t)f_hi << 32) | f_lo) & mask;
    const uint64_t rc = (((uint64_t)r_hi << 32) | r_lo) & mask;
    const uint32_t bm = funnel(st.B[sb / 32 + 1], st.B[sb / 32], sb & 31) & (K >= 32 ? 0xffffffffu : (1u << (K & 31)) - 1u);
    w = fw < rc ? fw : rc;
    return bm == 0;
}

template <int K, int J>
struct ForEachWindowW {
    template <class F>
    static __device__ __forceinline__ void run(const StreamsW &st, uint64_t lo, uint64_t hi, F &&f)
    {
        uint64_t w;
        const bool ok = window<K, J>(st, w) && w >= lo && w < hi;
        f(J, ok, w);
        ForEachWindowW<K, J + 1>::run(st, lo, hi, f);
    }
};
template <int K>
struct ForEachWindowW<K, KW_SEG> {
    template <class F>
    static __device__ __forceinline__ void run(const StreamsW &, uint64_t, uint64_t, F &&) {}
};

// one name for both widths: Windows<K>::Word, ::Streams, ::run(st, lo, hi, f)
template <int K, bool WIDE = (K > 16)>
struct Windows {
    typedef uint32_t Word;
    typedef Streams St;
    template <class F>
    static __device__ __forceinline__ void run(const St &st, Word lo, Word hi, F &&f) { ForEachWindow<K, 0>::run(st, lo, hi, f); }
};
template <int K>
struct Windows<K, true> {
    typedef uint64_t Word;
    typedef StreamsW St;
    template <class F>
    static __device__ __forceinline__ void run(const St &st, Word lo, Word hi, F &&f) { ForEachWindowW<K, 0>::run(st, lo, hi, f); }
};


}  // namespace
