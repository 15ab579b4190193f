// a4-a7: per-k-mer association scans over the bit-packed presence matrix.
//
// chi2_scan_kernel   replaces phenotypes.get_kmers_tested / conduct_chi_squared_test and helpers
//                    (modeling.py:677-714, :759-858)
// ttest_scan_kernel  replaces conduct_t_test / get_samples_distribution_for_ttest (:716-757)
//
// Layout: bits[M][wpr] u64, wpr even, so a row is wpr/2 16-byte chunks.  G = next power of two
// >= wpr/2 lanes own one row; every lane issues one 16-byte load per row (global_load_dwordx4,
// consecutive lanes -> consecutive addresses), popcounts its two words against the phenotype
// masks and the group combines with xor-shuffles.  A wave covers 64/G rows per step and keeps
// UNROLL steps of loads in flight.  HBM-read bound: 1 bit per k-mer x sample cell; no LDS, no MFMA.
// Up to 64 samples (r04; the reference's own example set has ~30) a row is ONE u64 (wpr = 1) and a lane's
// 16-byte load holds two rows: the kernels' G = 0 instantiations ("half a lane per row", 128 rows per wave
// step); masks and per-sample tables stay padded to a whole 16-byte chunk (cpr = 1).
//
// Exactness: the 2x2 table is integer (unit weights), and the statistic is evaluated with the
// reference's own operation order in IEEE double (this file is compiled with -ffp-contract=off),
// so round(chi2, 2) and "%.2E" % p come out string-identical.  The expensive exact evaluation only
// runs on rows that a division-free test T*(ad-bc)^2 >= thr*R1*R0*K1*K0*(1-1e-9) cannot rule out.
#include "dev_utils.h"
#include "psk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

namespace {

// tuning knobs (overridable at build time for A/B runs: make EXTRA=-DPSK_SC_UNROLL=...)
#ifndef PSK_SC_UNROLL
#define PSK_SC_UNROLL 4
#endif
#ifndef PSK_SC_GRID_MULT
#define PSK_SC_GRID_MULT 16
#endif
#ifndef PSK_SC_NT
#define PSK_SC_NT 1
#endif
constexpr int SC_THREADS = 256;
#ifndef PSK_LUT_THREADS
#define PSK_LUT_THREADS 1024
#endif
constexpr int SC_LUT_THREADS = PSK_LUT_THREADS;   // workgroup of the moment scans that keep their nibble tables in LDS (one per CU)
constexpr size_t SC_LUT_MAX_BYTES = 132 * 1024;
constexpr int SC_UNROLL = PSK_SC_UNROLL;
// Survivors are appended to SC_NSEG independent segments (segment = blockIdx % SC_NSEG), each with its
// own counter on its own 128-byte line: one shared counter serialises at ~11 ns per append (r01: a
// matrix with 1 % survivors ran 15x slower than the stream rate).
constexpr int SC_NSEG = 256;
constexpr int SC_CNT_STRIDE = 32;  // u32 per counter slot
constexpr int SC_INL_WORDS = 16;   // mask words carried inside ScanArgs
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ScanArgs {
    const u32x4 *bits;
    uint64_t M;
    int cpr;  // 16-byte chunks per row = wpr / 2 (1 when half)
    int half; // rows are ONE u64 (<= 64 samples): row r sits at byte 8 r; masks / tables as for one chunk
    // chi2
    const uint64_t *m1, *m0;   // phenotype == 1 / == 0 masks (wpr words each)
    const double *tab;         // per-sample table of the lane-per-row pass: [wpr*64][NM] doubles (see row_moments)
    const double *lut;         // the same table summed over every subset of each group of 4 samples (row_moments_lut)
    int c_lut;                 // ... for the first c_lut chunks of a row; the rest of the row takes the per-sample form
    const double *raw;         // Welch: {weight (0 for NA), phenotype value (0 for NA)} per sample, for the exact second pass
    const float *lut6;         // f32 six-bit table of the same moments (row_moments_f32) -- candidate selection only
    double e0, e1, e2;         // ... and what its sums may be off by: |sum w| <= e0, |sum w u| <= e1, |sum w u^2| <= e2 (chi2: e0 = class 1, e1 = class 0)
    double eref;               // Welch: what the REFERENCE's own arithmetic may be off by in a group mean (it sums the raw, unshifted values)
    int n1, n0;                // popcounts of the masks
    double W1, W0;             // weight totals of the two phenotype classes
    // t-test
    const uint64_t *mvalid;    // non-NA mask
    int nvalid;
    // filters
    int min_samples, max_samples;
    double pcut, pcut_bonf, thr;  // thr: statistic threshold of the division-free pre-test
    double tcrit;                 // t-test: |t| a row must exceed to be a candidate
    int omit_B;
    // output (SoA), counter
    uint64_t *res_row;
    double *res_stat, *res_p, *res_mx, *res_my;
    int32_t *res_nw;
    uint32_t *counter;   // SC_NSEG slots, SC_CNT_STRIDE u32 apart: [0] appended entries, [1] finished workgroups
    uint32_t seg_cap;    // entries per segment
    // end of a scan: the last workgroup of a segment (chi2) / the segment's finalize workgroup (Welch) publishes
    // the segment's count to final_counts (device, compact) and host_counts (pinned host memory, written
    // straight from the kernel) and zeroes the counter for the next scan -- no memset, no read-back copy
    uint32_t *final_counts, *host_counts;
    // phenotype masks of up to 1024 samples travel in the kernel arguments (no upload per scan)
    int inline_masks;
    uint64_t m1_inl[SC_INL_WORDS], m0_inl[SC_INL_WORDS];
};

__device__ __forceinline__ uint64_t reserve_slot(const ScanArgs &P)
{
    const uint32_t seg = blockIdx.x & (SC_NSEG - 1);
    const uint32_t idx = atomicAdd(&P.counter[seg * SC_CNT_STRIDE], 1u);
    return (uint64_t)seg * P.seg_cap + (idx < P.seg_cap ? idx : P.seg_cap - 1);
}

// Called by every thread at the very end of a chi2 scan workgroup: the LAST workgroup of a segment to get here
// publishes the segment's count and re-arms the counter (ticket = second word of the counter's 128-byte line).
__device__ __forceinline__ void publish_segment(const ScanArgs &P)
{
    // No fence: the count lives in device-scope atomics only, and every append of this workgroup has returned
    // its slot index (it was needed for the stores) before the barrier.  A __threadfence() here is an L2
    // write-back + invalidate per workgroup on this multi-XCD part and tripled the kernel time (r01).
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint32_t seg = blockIdx.x & (SC_NSEG - 1);
    const uint32_t n_blocks = (gridDim.x - seg + SC_NSEG - 1) / SC_NSEG;  // workgroups that map to this segment
    uint32_t *slot = &P.counter[seg * SC_CNT_STRIDE];
    if (atomicAdd(slot + 1, 1u) == n_blocks - 1) {
        const uint32_t c = atomicExch(slot, 0u);
        slot[1] = 0;
        P.final_counts[seg] = c;
        P.host_counts[seg] = c;
    }
}

// modeling.py:773-794 in the reference's operation order.
__device__ __forceinline__ double chi2_exact(double A, double B, double C, double D)
{
    const double w_pheno = A + B, wo_pheno = C + D, w_kmer = A + C, wo_kmer = B + D;
    const double total = w_pheno + wo_pheno;
    const double e0 = (w_pheno * w_kmer) / total, e1 = (w_pheno * wo_kmer) / total;
    const double e2 = (wo_pheno * w_kmer) / total, e3 = (wo_pheno * wo_kmer) / total;
    double stat = 0.0, d;
    d = A - e0; stat += (d * d) / e0;
    d = B - e1; stat += (d * d) / e1;
    d = C - e2; stat += (d * d) / e2;
    d = D - e3; stat += (d * d) / e3;
    return stat;
}


// ---- rows -----------------------------------------------------------------------------------------
// G = 0 stands for "half a lane per row" (8-byte rows, two per 16-byte load)
constexpr int sc_rpw(int G) { return G == 0 ? 128 : 64 / G; }   // rows per wave step
constexpr int sc_lanes(int G) { return G == 0 ? 1 : G; }        // lanes that share a load group
template <bool HALF>
__device__ __forceinline__ const u32x4 *sc_row_ptr(const ScanArgs &P, uint64_t r)
{
    if (HALF) return reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint2 *>(P.bits) + r);
    return P.bits + r * (uint64_t)P.cpr;
}
// chunk ch of the row at rp; an 8-byte row is its chunk 0 with an empty upper half
template <bool HALF>
__device__ __forceinline__ u32x4 sc_ld_chunk(const u32x4 *__restrict__ rp, int ch)
{
    if (HALF) {
        const uint2 v = *reinterpret_cast<const uint2 *>(rp);
        return (u32x4){v.x, v.y, 0u, 0u};
    }
    return rp[ch];
}

// ---- lane-per-row moments -----------------------------------------------------------------------
// Rows that pass the popcount frequency filter need f64 sums over their present samples (class weight
// sums for the weighted chi2, weighted moments for Welch).  They are queued per wave and handled 64 at a
// time, ONE ROW PER LANE: every lane walks its own row while all lanes visit the same sample s at the
// same time, so the per-sample table entries tab[s][0..NM) are wave-uniform and come through the scalar
// data cache into SGPRs (constant address space => s_load), not through LDS or the vector pipe.  A cell
// costs 2 + NM VALU ops: the presence bit becomes 0.0 / 1.0 (v_bfe_i32 + v_and 0x3FF00000 on the high
// word), then one v_fma_f64 per moment with the table entry as an SGPR operand -- no cross-lane
// reduction at all.  The sums associate differently from the reference's sample-order loops (two interleaved
// accumulators here, groups of four samples in the table form below): ~1e-15 relative from the reference.  The Welch
// statistics are used as they come (compared at 1e-8); the weighted chi2 uses these sums for its pre-test only and
// re-sums the candidates in the reference's order (chi2w_finalize_kernel; DESIGN.md "Exactness strategy").
// (r01: the previous whole-wave-per-row form spent ~1000 cycles per row in LDS latency and three DPP wave
// sums: 11.1 ms for 16 M x 1024 with a third of the rows passing.)
typedef const __attribute__((address_space(4))) double *cdptr;
// queue entries per wave: < 64 carried over + <= 64 / G appended per step of an unrolled batch
constexpr int rq_cap(int G, int unroll = SC_UNROLL) { return 64 + (G == 0 ? 128 : 64 / G) * unroll; }
#ifndef PSK_LUT_UNROLL
#define PSK_LUT_UNROLL 8
#endif
#ifndef PSK_LUT_NT
#define PSK_LUT_NT 1     // streaming loads of the table-in-LDS kernels carry the nontemporal hint
#endif
// rows in flight per lane group of the table-in-LDS kernels (half the waves per CU of the plain ones); fewer where a
// wave step covers many rows, so that the waves' queues stay small beside the table
constexpr int lut_unroll(int G) { return G == 0 ? (PSK_LUT_UNROLL < 2 ? PSK_LUT_UNROLL : 2) : G == 1 ? (PSK_LUT_UNROLL < 4 ? PSK_LUT_UNROLL : 4) : G == 2 ? (PSK_LUT_UNROLL < 8 ? PSK_LUT_UNROLL : 8) : PSK_LUT_UNROLL; }

template <int NM, bool HALF = false>
__device__ __forceinline__ void row_moments(const u32x4 *__restrict__ rp, int cpr, cdptr tab, double *acc)
{
    double a0[NM], a1[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) { a0[m] = 0.0; a1[m] = 0.0; }
    u32x4 y = sc_ld_chunk<HALF>(rp, 0);
    for (int ch = 0; ch < cpr; ch++) {
        const uint32_t w4[4] = {y.x, y.y, y.z, y.w};
        if (ch + 1 < cpr) y = rp[ch + 1];
        cdptr tp = tab + (size_t)ch * 128 * NM;
#pragma unroll
        for (int h = 0; h < (HALF ? 2 : 4); h++) {
#pragma unroll
            for (int b = 0; b < 32; b += 2) {
                const uint32_t h0 = (uint32_t)(((int32_t)(w4[h] << (31 - b))) >> 31) & 0x3FF00000u;
                const uint32_t h1 = (uint32_t)(((int32_t)(w4[h] << (30 - b))) >> 31) & 0x3FF00000u;
                const double f0 = __hiloint2double((int)h0, 0), f1 = __hiloint2double((int)h1, 0);
#pragma unroll
                for (int m = 0; m < NM; m++) {
                    a0[m] = fma(f0, tp[(h * 32 + b) * NM + m], a0[m]);
                    a1[m] = fma(f1, tp[(h * 32 + b + 1) * NM + m], a1[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NM; m++) acc[m] = a0[m] + a1[m];
}

// The same sums from a nibble table: lut[g][p][0..NM) = sum of tab[4 g + b] over the bits b set in p (ascending b), for
// every group g of 4 samples and every 4-bit pattern p, held in LDS.  A lane then spends one nibble extract, one
// address and ONE LDS read + NM adds per FOUR samples instead of (2 + NM) VALU instructions per sample: the lanes of
// a wave (64 different rows) look up the same group at the same time, so their 16 possible addresses are 16 x NM x 8
// consecutive bytes -- for NM = 2 exactly the 64 banks, without a conflict; equal patterns are broadcast.
// (r01: the per-sample form was f64-VALU bound, 0.76 ms for 16 M x 1024 with 20 % of the rows passing.)
// The sums associate differently from the reference's sample-order loops: a group's members are added first, then
// the groups in order (two interleaved accumulators, as before) -- 1e-15 relative, see DESIGN.md "Exactness".
template <int NM>
__global__ void moment_lut_kernel(const double *__restrict__ tab, int n_groups, double *__restrict__ lut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * 16) return;
    const int g = i >> 4, p = i & 15;
#pragma unroll
    for (int m = 0; m < NM; m++) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if ((p >> b) & 1) s += tab[(size_t)(4 * g + b) * NM + m];
        // NM = 3: the first two moments as 16-byte pairs, the third in a table of its own behind them (row_moments_lut)
        if (NM == 3) lut[m < 2 ? (size_t)i * 2 + m : (size_t)n_groups * 32 + i] = s;
        else lut[(size_t)i * NM + m] = s;
    }
}

// The row itself is read SC_LUT_PF chunks at a time, all loads issued before the first lookup: read one chunk ahead
// (r02 at first) every chunk paid a global-load latency of its own, and THAT, not the LDS pipe, set the time of the pass
// (~8 us per 64 rows of 1024 samples against 1.7 us of lookups).
constexpr int SC_LUT_PF = 8;
template <int NM, bool HALF = false>
__device__ __forceinline__ void row_moments_lut(const u32x4 *__restrict__ rp, int cpr, const double *lut, double *acc)
{
    double a0[NM], a1[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) { a0[m] = 0.0; a1[m] = 0.0; }
    for (int c0 = 0; c0 < cpr; c0 += SC_LUT_PF) {
        u32x4 y[SC_LUT_PF];
#pragma unroll
        for (int i = 0; i < SC_LUT_PF; i++) y[i] = c0 + i < cpr ? sc_ld_chunk<HALF>(rp, c0 + i) : (u32x4)(0u);
#pragma unroll
        for (int i = 0; i < SC_LUT_PF; i++) {
            if (c0 + i >= cpr) break;
            const uint32_t w4[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
            // 32 groups of 4 samples per 16-byte chunk.  NM = 3: entries of 24 bytes were read as ds_read2_b64 + ds_read_b64
            // (8 + 2 LDS cycles, banks mod 32); pairs {m0, m1} and a separate table of m2 are a ds_read_b128 and a
            // ds_read_b64 (4 + 2 cycles, both conflict-free: 16 entries = 64 resp. 32 of the 64 banks)
            const double *lp = lut + (size_t)(c0 + i) * 32 * 16 * (NM == 3 ? 2 : NM);
            const double *lp2 = lut + (size_t)cpr * 32 * 16 * 2 + (size_t)(c0 + i) * 32 * 16;   // NM = 3 only
#pragma unroll
            for (int h = 0; h < (HALF ? 2 : 4); h++) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const uint32_t i0 = (h * 8 + k) * 16 + ((w4[h] >> (4 * k)) & 15u), i1 = (h * 8 + k + 1) * 16 + ((w4[h] >> (4 * k + 4)) & 15u);
                    const double *e0 = lp + i0 * (NM == 3 ? 2 : NM);
                    const double *e1 = lp + i1 * (NM == 3 ? 2 : NM);
                    if (NM == 3) {
                        const double2 v0 = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(e0, 16));
                        const double2 v1 = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(e1, 16));
                        a0[0] += v0.x; a0[1] += v0.y; a1[0] += v1.x; a1[1] += v1.y;
                        a0[NM - 1] += lp2[i0]; a1[NM - 1] += lp2[i1];
                    } else if (NM == 2) {
                        // ONE 16-byte read per entry (ds_read_b128: 4 LDS cycles, banks mod 64, the 16 entries of a
                        // group = the 64 banks).  Read as two doubles it became ds_read2_b64 -- 8 cycles, banks mod 32,
                        // every group 2-way conflicted: 41 % of the LDS cycles of the pass (SQ_LDS_BANK_CONFLICT, r02)
                        const double2 v0 = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(e0, 16));
                        const double2 v1 = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(e1, 16));
                        a0[0] += v0.x; a0[NM - 1] += v0.y; a1[0] += v1.x; a1[NM - 1] += v1.y;
                    } else {
#pragma unroll
                        for (int m = 0; m < NM; m++) { a0[m] += e0[m]; a1[m] += e1[m]; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NM; m++) acc[m] = a0[m] + a1[m];
}

// ---- six-bit tables in f32: candidate selection at a third of the cost ------------------------------------------------
// Since r03 every moment scan decides in a second kernel that re-sums its candidates exactly (chi2w_finalize_kernel,
// ttest_finalize_kernel), so the sums of the streaming kernel only have to be good enough to not MISS a candidate.  They are
// therefore taken in f32 from a table over groups of SIX samples: a 16-byte chunk of a row is 21 six-bit groups + one
// two-bit group, i.e. 22 lookups instead of 32; an entry of two moments is 8 bytes (one ds_read_b64, 4 LDS cycles
// instead of 8) and is accumulated by ONE v_pk_add_f32 (4 VALU cycles instead of two v_add_f64 = 16).  The kernel turns
// the f32 sums into an UPPER bound of the statistic with the rounding-error bounds the host derives from the table
// itself (e0, e1, e2: (additions per accumulator + 3) x 2^-24 x the sum of the absolute terms over all samples, which
// bounds the error of any subset's f32 sum), and every row whose bound reaches the threshold is a candidate.
// Layout: per chunk 21 x 64 + 4 = SC_L6_ENTRIES entries; float2 {m0, m1} per entry, then -- three moments -- one float
// per entry in a second table behind the first.  1,024 samples: 86 KB (two moments), 129 KB (three).
constexpr int SC_L6_ENTRIES = 21 * 64 + 4;
__host__ __device__ inline size_t lut6_bytes(int chunks, int nm) { return (size_t)chunks * SC_L6_ENTRIES * (nm == 3 ? 12 : 8); }

template <int NM>
__global__ void moment_lut6_kernel(const double *__restrict__ tab, int chunks, float *__restrict__ lut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunks * SC_L6_ENTRIES) return;
    const int ch = i / SC_L6_ENTRIES, e = i % SC_L6_ENTRIES;
    const int j = e < 21 * 64 ? e >> 6 : 21, p = e < 21 * 64 ? e & 63 : e - 21 * 64, width = j < 21 ? 6 : 2;
    const size_t s0 = (size_t)ch * 128 + 6 * j;
#pragma unroll
    for (int m = 0; m < NM; m++) {
        double s = 0.0;
        for (int b = 0; b < width; b++)
            if ((p >> b) & 1) s += tab[(s0 + b) * NM + m];
        if (m < 2) lut[(size_t)i * 2 + m] = (float)s;
        else lut[(size_t)chunks * SC_L6_ENTRIES * 2 + i] = (float)s;
    }
}

typedef float sc_f32x2 __attribute__((ext_vector_type(2)));

// f32 sums of NM moments over the present samples of one row (one row per lane), from the six-bit tables in LDS
template <int NM, bool HALF = false>
__device__ __forceinline__ void row_moments_f32(const u32x4 *__restrict__ rp, int cpr, const float *lut, double *acc)
{
    sc_f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float c0 = 0.f, c1 = 0.f;
    const float *lut3 = lut + (size_t)cpr * SC_L6_ENTRIES * 2;   // NM = 3 only
    for (int g0 = 0; g0 < cpr; g0 += SC_LUT_PF) {
        u32x4 y[SC_LUT_PF];
#pragma unroll
        for (int i = 0; i < SC_LUT_PF; i++) y[i] = g0 + i < cpr ? sc_ld_chunk<HALF>(rp, g0 + i) : (u32x4)(0u);
#pragma unroll
        for (int i = 0; i < SC_LUT_PF; i++) {
            if (g0 + i >= cpr) break;
            const uint32_t w4[5] = {y[i].x, y[i].y, y[i].z, y[i].w, 0u};
            const sc_f32x2 *lp = reinterpret_cast<const sc_f32x2 *>(lut) + (size_t)(g0 + i) * SC_L6_ENTRIES;
            const float *lp3 = lut3 + (size_t)(g0 + i) * SC_L6_ENTRIES;
#pragma unroll
            for (int j = 0; j < (HALF ? 11 : 22); j++) {   // an 8-byte row: samples 0 ... 63 lie in groups 0 ... 10
                const int o = 6 * j, wi = o >> 5, sh = o & 31;
                uint32_t idx;
                if (j == 21) idx = w4[3] >> 30;
                else if (sh <= 26) idx = (w4[wi] >> sh) & 63u;
                else idx = __builtin_amdgcn_alignbit(w4[wi + 1], w4[wi], sh) & 63u;
                const uint32_t e = (uint32_t)j * 64u + idx;
                const sc_f32x2 v = lp[e];
                if (j & 1) a1 += v; else a0 += v;
                if (NM == 3) { if (j & 1) c1 += lp3[e]; else c0 += lp3[e]; }
            }
        }
    }
    const sc_f32x2 a = a0 + a1;
    acc[0] = (double)a.x;
    acc[1] = (double)a.y;
    if (NM == 3) acc[2] = (double)(c0 + c1);
}

// Both forms in one row, for rows whose table does not fit the LDS: the first c_lut chunks through the nibble table, the
// others per sample.  (Splitting a row that does fit in halves, to keep the LDS pipe and the f64 VALU busy at the same
// time, did not pay: 16 M x 1024 with a fifth of the rows passing took 0.66 ms against 0.63 ms with the whole row in
// the table and 0.75 ms per sample, r02.  8 M x 2048, where half the row fits: 0.67 ms against 0.95 ms per sample.)
template <int NM, bool HALF = false>
__device__ __forceinline__ void row_moments_mixed(const u32x4 *__restrict__ rp, int cpr, int c_lut, const double *lut, cdptr tab,
                                                  double *acc)
{
    double a[NM], b[NM];
    row_moments_lut<NM, HALF>(rp, c_lut, lut, a);
#pragma unroll
    for (int m = 0; m < NM; m++) b[m] = 0.0;
    if (c_lut < cpr) row_moments<NM, HALF>(rp + c_lut, cpr - c_lut, tab + (size_t)c_lut * 128 * NM, b);   // (half: c_lut is 0 or 1 = cpr)
#pragma unroll
    for (int m = 0; m < NM; m++) acc[m] = a[m] + b[m];
}

// the workgroup's copy of the nibble table: global -> LDS, 16 bytes per thread and step
__device__ __forceinline__ void load_lut(double *lds, const double *__restrict__ g, int n_doubles, int threads)
{
    const double2 *src = reinterpret_cast<const double2 *>(g);
    double2 *dst = reinterpret_cast<double2 *>(lds);
    for (int i = threadIdx.x; i < n_doubles / 2; i += threads) dst[i] = src[i];
    __syncthreads();
}

// appends the rows flagged in this step (one flag per lane group leader) to the wave's queue
__device__ __forceinline__ int queue_rows(bool flag, uint64_t row, int2 v, uint64_t *q_row, int2 *q_val, int q, int lane)
{
    const uint64_t todo = __ballot(flag);
    if (!todo) return q;
    if (flag) {
        const int pos = q + __popcll(todo & ((1ull << lane) - 1ull));
        q_row[pos] = row;
        q_val[pos] = v;
    }
    return __builtin_amdgcn_readfirstlane(q + __popcll(todo));  // keep the count in an SGPR
}

// drops the first 64 entries of the wave's queue (the rest moves down 64 places, 64 entries at a time)
__device__ __forceinline__ int queue_pop64(uint64_t *q_row, int2 *q_val, int q, int lane)
{
    const int rest = q - 64;
    for (int base = 0; base < rest; base += 64) {
        uint64_t r = 0;
        int2 n = make_int2(0, 0);
        const bool mv = base + lane < rest;
        if (mv) { r = q_row[64 + base + lane]; n = q_val[64 + base + lane]; }
        __builtin_amdgcn_wave_barrier();
        if (mv) { q_row[base + lane] = r; q_val[base + lane] = n; }
        __builtin_amdgcn_wave_barrier();
    }
    return __builtin_amdgcn_readfirstlane(rest > 0 ? rest : 0);
}

// MODE 0: unit weights, the exact evaluation in line -- the usual case, where almost no row passes the pre-test.
// MODE 1: GSC weights: rows that pass the frequency filter are queued per wave and handled 64 at a time, one per lane.
// MODE 2: unit weights, rows that pass the pre-test are queued the same way: a scan with many survivors
//         (--omit_B_correction keeps ~pvalue of all rows) then evaluates 64 of them per pass instead of one or two
//         lanes of a wave at a time.  Same formulas, same results as MODE 0 (the host picks, see pick_chi2_mode).
template <int G, int MODE, bool LUT = false, bool F32 = false>
__global__ __launch_bounds__(LUT ? SC_LUT_THREADS : SC_THREADS) void chi2_scan_kernel(const ScanArgs P)
{
    constexpr bool WEIGHTED = MODE == 1, QUEUED = MODE != 0;
    constexpr bool HALF = G == 0;          // 8-byte rows, two per load
    constexpr int GL = sc_lanes(G), NSUB = HALF ? 2 : 1;
    constexpr int THREADS = LUT ? SC_LUT_THREADS : SC_THREADS;
    constexpr int UNR = LUT ? lut_unroll(G) : SC_UNROLL;
    __shared__ uint64_t s_qrow[QUEUED ? THREADS / 64 : 1][QUEUED ? rq_cap(G, UNR) : 1];
    __shared__ int2 s_qval[QUEUED ? THREADS / 64 : 1][QUEUED ? rq_cap(G, UNR) : 1];
    extern __shared__ __attribute__((aligned(16))) double s_lut[];   // LUT: the nibble table of row_moments_lut / the six-bit f32 table
    if (LUT && F32) load_lut(s_lut, reinterpret_cast<const double *>(P.lut6), (int)(lut6_bytes(P.cpr, 2) / 8), THREADS);
    else if (LUT) load_lut(s_lut, P.lut, P.c_lut * 32 * 16 * 2, THREADS);
    constexpr int RPW = sc_rpw(G);  // rows per wave step
    const int lane = threadIdx.x & 63;
    const int g = lane & (GL - 1);
    const int rsub = HALF ? 2 * lane : lane / GL;
    const uint64_t n_steps = (P.M + RPW - 1) / RPW;
    const uint64_t wave_global = (uint64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    const uint64_t total_waves = (uint64_t)gridDim.x * (THREADS / 64);
    const bool has_chunk = g < P.cpr;
    uint64_t m1a = 0, m1b = 0, m0a = 0, m0b = 0;
    if (has_chunk) {
        if (P.inline_masks) { m1a = P.m1_inl[2 * g]; m1b = P.m1_inl[2 * g + 1]; m0a = P.m0_inl[2 * g]; m0b = P.m0_inl[2 * g + 1]; }
        else { m1a = P.m1[2 * g]; m1b = P.m1[2 * g + 1]; m0a = P.m0[2 * g]; m0b = P.m0[2 * g + 1]; }
    }
    uint64_t *q_row = s_qrow[QUEUED ? (threadIdx.x >> 6) : 0];
    int2 *q_val = s_qval[QUEUED ? (threadIdx.x >> 6) : 0];
    int q = 0;
    // `cnt` queued rows, one per lane.  Weighted: class weight sums in sample order, then the same pre-test / exact
    // statistic / keep rule as the unweighted path.  MODE 2: the row has passed the pre-test; (a, c) came with it.
    // (Queueing costs the usual sparse case 4 % -- r01 A/B on cfg 2: 110.7 vs 115.2 us -- hence MODE 0.)
    auto process = [&](int cnt) {
        const bool act = lane < cnt;
        const uint64_t r = q_row[act ? lane : 0];
        const int2 qv = q_val[act ? lane : 0];
        int r_nw = qv.x;
        double A, B, C, D;
        if (WEIGHTED && F32) {
            // f32 class-weight sums (A, C may be off by e0, e1): an UPPER bound of the statistic decides who is a candidate.
            // det = AD - BC = A W0 - C W1 is linear in the two sums; the column totals shrink by the error
            double ws[2];
            row_moments_f32<2, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, reinterpret_cast<const float *>(s_lut), ws);
            const double T = P.W1 + P.W0, err = P.e0 + P.e1;
            const double det = fabs(ws[0] * P.W0 - ws[1] * P.W1) + (P.e0 * P.W0 + P.e1 * P.W1);
            const double K1 = (ws[0] + ws[1]) - err, K0 = (T - (ws[0] + ws[1])) - err;
            const bool cand = !(K1 > 0.0 && K0 > 0.0) || !(T * det * det < P.thr * P.W1 * P.W0 * K1 * K0 * (1.0 - 1e-9));
            if (act && cand) {
                const uint64_t idx = reserve_slot(P);
                P.res_row[idx] = r;
                P.res_nw[idx] = r_nw;
            }
            return;
        } else if (WEIGHTED) {
            double ws[2];
            if (LUT) row_moments_mixed<2, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, P.c_lut, s_lut, (cdptr)P.tab, ws);
            else row_moments<2, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, (cdptr)P.tab, ws);
            A = ws[0]; B = P.W1 - ws[0]; C = ws[1]; D = P.W0 - ws[1];
            const double R1 = A + B, R0 = C + D, K1 = A + C, K0 = B + D, T = R1 + R0;
            const double det = A * D - B * C;
            const double lhs = T * det * det, rhs = P.thr * R1 * R0 * K1 * K0;
            // candidates only: chi2w_finalize_kernel gives them the reference's own cells and decides (the sums here
            // associate differently, ~1e-15: hence the 1e-9 margin)
            if (act && !(lhs < rhs * (1.0 - 1e-9))) {
                const uint64_t idx = reserve_slot(P);
                P.res_row[idx] = r;
                P.res_nw[idx] = r_nw;
            }
            return;
        } else {
            if (!act) return;
            A = (double)qv.x; B = (double)(P.n1 - qv.x); C = (double)qv.y; D = (double)(P.n0 - qv.y);
            r_nw = qv.x + qv.y;
        }
        const double stat = chi2_exact(A, B, C, D);
        const double p = exp(-0.5 * stat);
        const bool keep = (P.omit_B && p < P.pcut) || (p < P.pcut_bonf);
        if (keep) {
            const uint64_t idx = reserve_slot(P);
            P.res_row[idx] = r;
            P.res_stat[idx] = stat;
            P.res_p[idx] = p;
            P.res_nw[idx] = r_nw;
        }
    };

    // one copy of process(): the queue is drained after each unrolled batch and, once the rows run out,
    // down to empty (keeps its registers and code out of the streaming part)
    for (uint64_t s0 = wave_global * UNR;; s0 += total_waves * UNR) {
        const bool more = s0 < n_steps;
        if (QUEUED) {
            while (q >= 64 || (!more && q > 0)) {
                process(q < 64 ? q : 64);
                q = queue_pop64(q_row, q_val, q, lane);
            }
        }
        if (!more) break;
        u32x4 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint64_t row = (s0 + u) * RPW + rsub;
            x[u] = (u32x4)(0u);
            if (HALF) {   // rows `row` and `row + 1` in one 16-byte load (the matrix starts 16-byte aligned and `row` is even)
                const u32x4 *pp = reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint2 *>(P.bits) + row);
                if (row + 1 < P.M) x[u] = __builtin_nontemporal_load(pp);
                else if (row < P.M) { const uint2 v = *reinterpret_cast<const uint2 *>(pp); x[u].x = v.x; x[u].y = v.y; }   // the odd last row
            } else if (row < P.M && has_chunk) {
#if PSK_SC_NT
                if (LUT && !PSK_LUT_NT) x[u] = P.bits[row * (uint64_t)P.cpr + g];
                else x[u] = __builtin_nontemporal_load(&P.bits[row * (uint64_t)P.cpr + g]);
#else
                x[u] = P.bits[row * (uint64_t)P.cpr + g];
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
          for (int sub = 0; sub < NSUB; sub++) {
            const uint64_t row = (s0 + u) * RPW + rsub + sub;
            const uint64_t xa = sub ? (((uint64_t)x[u].w << 32) | x[u].z) : (((uint64_t)x[u].y << 32) | x[u].x);
            const uint64_t xb = HALF ? 0ull : (((uint64_t)x[u].w << 32) | x[u].z);
            uint32_t a = __popcll(xa & m1a) + (HALF ? 0u : (uint32_t)__popcll(xb & m1b));
            uint32_t c = __popcll(xa & m0a) + (HALF ? 0u : (uint32_t)__popcll(xb & m0b));
            if (!HALF && P.cpr > GL) {  // rows wider than 64 chunks (more than 8192 samples)
                if (row < P.M)
                    for (int ch = g + GL; ch < P.cpr; ch += GL) {
                        const u32x4 y = P.bits[row * (uint64_t)P.cpr + ch];
                        const uint64_t ya = ((uint64_t)y.y << 32) | y.x, yb = ((uint64_t)y.w << 32) | y.z;
                        a += __popcll(ya & P.m1[2 * ch]) + __popcll(yb & P.m1[2 * ch + 1]);
                        c += __popcll(ya & P.m0[2 * ch]) + __popcll(yb & P.m0[2 * ch + 1]);
                    }
            }
#pragma unroll
            for (int d = GL / 2; d > 0; d >>= 1) {
                a += __shfl_xor(a, d, 64);
                c += __shfl_xor(c, d, 64);
            }
            const int n_w = (int)(a + c);
            const int n_wo = (P.n1 - (int)a) + (P.n0 - (int)c);
            const bool freq_ok = (row < P.M) && !(n_w < P.min_samples || n_wo < 2 || n_w > P.max_samples);
            if (WEIGHTED) {
                q = queue_rows(freq_ok && g == 0, row, make_int2(n_w, 0), q_row, q_val, q, lane);
                continue;
            }
            const double A = (double)a, B = (double)(P.n1 - (int)a), C = (double)c, D = (double)(P.n0 - (int)c);
            // division-free pre-test: chi2 = T (AD - BC)^2 / (R1 R0 K1 K0)
            const double R1 = A + B, R0 = C + D, K1 = A + C, K0 = B + D, T = R1 + R0;
            const double det = A * D - B * C;
            const double lhs = T * det * det, rhs = P.thr * R1 * R0 * K1 * K0;
            const bool cand = freq_ok && g == 0 && !(lhs < rhs * (1.0 - 1e-9));  // NaN compares false -> a candidate
            if (MODE == 2) {
                q = queue_rows(cand, row, make_int2((int)a, (int)c), q_row, q_val, q, lane);
                continue;
            }
            if (!cand) continue;
            const double stat = chi2_exact(A, B, C, D);
            const double p = exp(-0.5 * stat);  // chi2.sf(stat, df = 2), modeling.py:782-792
            const bool keep = (P.omit_B && p < P.pcut) || (p < P.pcut_bonf);  // modeling.py:795
            if (keep) {
                const uint64_t idx = reserve_slot(P);
                P.res_row[idx] = row;
                P.res_stat[idx] = stat;
                P.res_p[idx] = p;
                P.res_nw[idx] = n_w;
            }
        }
    }
    if (!WEIGHTED) publish_segment(P);   // weighted: chi2w_finalize_kernel publishes
}

// Second pass of the weighted chi2 scan: one workgroup per result segment, one candidate per lane.  The 2 x 2 table is
// summed again exactly as the reference does it (modeling.py:809-823: every sample in order adds its weight to ONE of
// the four cells; here the other three get + 0.0, and a weight times 1.0 or 0.0 is exact, so one fma per cell IS that
// addition), so the statistic, round(chi2, 2) and "%.2E" of the
// p-value are the reference's to the last bit; then the keep rule of modeling.py:795 and the compaction of the segment
// in place.  (Doing this inside the scan kernel, per 64 queued rows with at least one candidate, cost 2.3 ms instead of
// 0.6 ms for 16 M x 1024: the sample-order sums are a dependent chain of 1024 f64 adds whatever the number of live lanes.)
constexpr int SC_FIN_THREADS = 1024;
constexpr int SC_FIN_BLK = 16;         // chunks (of 128 samples) of the weight table staged in LDS at a time: 32 KB
__global__ __launch_bounds__(SC_FIN_THREADS) void chi2w_finalize_kernel(const ScanArgs P)
{
    __shared__ uint32_t scan_lds[SC_FIN_THREADS / 64];
    __shared__ uint32_t s_out;
    // {weight if phenotype 1, weight if phenotype 0} of the samples of the current block: every lane reads the SAME
    // entry at the same time (one broadcast ds_read_b128).  As scalar loads from global memory the entries cost
    // ~330 cycles per sample (0.14 ms for 160 k candidates of 1024 samples); from LDS the pass is its 4 + 3 VALU
    // operations per sample.
    __shared__ double2 s_tab[SC_FIN_BLK * 128];
    const uint32_t seg = blockIdx.x;
    const uint32_t c = P.counter[seg * SC_CNT_STRIDE];
    const uint64_t base = (uint64_t)seg * P.seg_cap;
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    for (uint32_t s0 = 0; s0 < c; s0 += SC_FIN_THREADS) {
        const uint32_t i = s0 + threadIdx.x;
        const bool valid = i < c;
        const uint64_t row = valid ? P.res_row[base + i] : 0;
        const int32_t nw = valid ? P.res_nw[base + i] : 0;
        const bool wave_any = __any(valid);
        const u32x4 *rp = P.half ? sc_row_ptr<true>(P, row) : sc_row_ptr<false>(P, row);
        // the four cells, sample by sample
        double ca = 0.0, cb = 0.0, cc = 0.0, cd = 0.0;
        for (int c0 = 0; c0 < P.cpr; c0 += SC_FIN_BLK) {
            const int nc = P.cpr - c0 < SC_FIN_BLK ? P.cpr - c0 : SC_FIN_BLK;
            __syncthreads();   // the previous block has been consumed
            for (int e = threadIdx.x; e < nc * 128; e += SC_FIN_THREADS)
                s_tab[e] = reinterpret_cast<const double2 *>(P.tab)[(size_t)c0 * 128 + e];
            __syncthreads();
            if (!wave_any) continue;
            for (int ch = 0; ch < nc; ch++) {
                const u32x4 y = P.half ? sc_ld_chunk<true>(rp, 0) : rp[c0 + ch];   // (an 8-byte row: samples 64 ... 127 have zero weights)
                const uint32_t w4[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int h = 0; h < 4; h++) {
#pragma unroll 8
                    for (int sb = 0; sb < 32; sb++) {
                        const uint32_t fh = (uint32_t)(((int32_t)(w4[h] << (31 - sb))) >> 31) & 0x3FF00000u;
                        const double f = __hiloint2double((int)fh, 0), g = __hiloint2double((int)(fh ^ 0x3FF00000u), 0);
                        const double2 t = s_tab[ch * 128 + h * 32 + sb];
                        ca = fma(f, t.x, ca);
                        cb = fma(g, t.x, cb);
                        cc = fma(f, t.y, cc);
                        cd = fma(g, t.y, cd);
                    }
                }
            }
        }
        double stat = 0.0, p = 1.0;
        bool keep = false;
        if (wave_any) {
            stat = chi2_exact(ca, cb, cc, cd);
            p = exp(-0.5 * stat);
            keep = valid && ((P.omit_B && p < P.pcut) || (p < P.pcut_bonf));
        }
        uint32_t tot;
        const uint32_t pos = psk_block_excl_scan_u32<SC_FIN_THREADS>(keep ? 1u : 0u, &tot, scan_lds);  // barriers inside
        const uint32_t out = s_out;
        if (keep) {
            const uint64_t o = base + out + pos;  // <= base + i: compaction only moves entries down
            P.res_row[o] = row; P.res_stat[o] = stat; P.res_p[o] = p; P.res_nw[o] = nw;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_out = out + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        P.counter[seg * SC_CNT_STRIDE] = 0;  // re-armed for the next scan
        P.final_counts[seg] = s_out;
        P.host_counts[seg] = s_out;
    }
}

// ---- Student-t two-sided p-value: I_{df/(df+t^2)}(df/2, 1/2), Lentz continued fraction ---------
__device__ double dev_betacf(double a, double b, double x)
{
    const double TINY = 1e-300, EPS = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < TINY) d = TINY;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 10000; m++) {
        const int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < EPS) break;
    }
    return h;
}

__device__ double dev_betainc(double a, double b, double x)
{
    if (!(x > 0.0)) return (x == 0.0) ? 0.0 : NAN;
    if (!(x < 1.0)) return (x == 1.0) ? 1.0 : NAN;
    const double lbt = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x);
    const double bt = exp(lbt);
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * dev_betacf(a, b, x) / a;
    return 1.0 - bt * dev_betacf(b, a, 1.0 - x) / b;
}

__device__ __attribute__((noinline)) double dev_t_two_sided_p(double t, double df)
{
    if (isnan(t) || isnan(df) || !(df > 0)) return NAN;
    if (isinf(t)) return 0.0;
    return dev_betainc(0.5 * df, 0.5, df / (df + t * t));
}

// Welch scan.  Phase A is the chi2 kernel's streaming shape (one 16-byte load per lane per row, popcount
// against the non-NA mask, group reduce, frequency filter of modeling.py:731).  Rows that pass are queued
// per wave and handled 64 at a time, one row per lane (row_moments): ONE pass of (weighted) moments of the
// k-mer-present group over phenotype values shifted by their global weighted mean -- the absent group
// follows from the totals -- then means, variances, t and the Satterthwaite df in the same lane.
// Table layout: unit weights tab[s] = {u, u*u} (n comes from the popcount); GSC weights {w, w*u, w*u*u};
// zeros for NA samples and padding.
template <int G, bool WT, bool LUT = false, bool F32 = false>
__global__ __launch_bounds__(LUT ? SC_LUT_THREADS : SC_THREADS) void ttest_scan_kernel(const ScanArgs P, const double mu)
{
    constexpr bool HALF = G == 0;          // 8-byte rows, two per load
    constexpr int GL = sc_lanes(G), NSUB = HALF ? 2 : 1;
    constexpr int THREADS = LUT ? SC_LUT_THREADS : SC_THREADS;
    constexpr int UNR = LUT ? lut_unroll(G) : SC_UNROLL;
    __shared__ uint64_t s_qrow[THREADS / 64][rq_cap(G, UNR)];
    __shared__ int2 s_qval[THREADS / 64][rq_cap(G, UNR)];
    constexpr int RPW = sc_rpw(G);
    constexpr int NM = WT ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) double s_lut[];   // LUT: the nibble table of row_moments_lut / the six-bit f32 table
    if (LUT && F32) load_lut(s_lut, reinterpret_cast<const double *>(P.lut6), (int)(lut6_bytes(P.cpr, NM) / 8), THREADS);
    else if (LUT) load_lut(s_lut, P.lut, P.c_lut * 32 * 16 * NM, THREADS);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane & (GL - 1);
    const int rsub = HALF ? 2 * lane : lane / GL;
    const uint64_t n_steps = (P.M + RPW - 1) / RPW;
    const uint64_t wave_global = (uint64_t)blockIdx.x * (THREADS / 64) + wid;
    const uint64_t total_waves = (uint64_t)gridDim.x * (THREADS / 64);
    const bool has_chunk = g < P.cpr;
    uint64_t mva = 0, mvb = 0;
    if (has_chunk) { mva = P.mvalid[2 * g]; mvb = P.mvalid[2 * g + 1]; }
    uint64_t *q_row = s_qrow[wid];
    int2 *q_val = s_qval[wid];
    int q = 0;  // queued rows (wave-uniform)

    auto process = [&](int cnt) {
        const bool act = lane < cnt;
        const uint64_t r = q_row[act ? lane : 0];
        const int r_nw = q_val[act ? lane : 0].x;
        double mo[NM];
        if (F32) row_moments_f32<NM, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, reinterpret_cast<const float *>(s_lut), mo);
        else if (LUT) row_moments_mixed<NM, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, P.c_lut, s_lut, (cdptr)P.tab, mo);
        else row_moments<NM, HALF>(sc_row_ptr<HALF>(P, r), P.cpr, (cdptr)P.tab, mo);
        if (!act) return;
        const double nx = WT ? mo[0] : (double)r_nw, sx = mo[NM - 2], qx = mo[NM - 1];
        const double ny = P.W1 - nx, sy = P.W0 - sx, qy = P.thr - qx;  // totals: W1 = sum w, W0 = sum w*u, thr = sum w*u^2
        if (F32) {
            // the sums are f32 sums, off by at most e0 (sum w; 0 with unit weights: the popcount), e1 (sum w u), e2 (sum w u^2):
            // an UPPER bound of |t| -- the largest difference of the means over the smallest standard error the bounds
            // allow -- decides who is a candidate (ttest_finalize_kernel computes the statistic itself)
            const double e0 = WT ? P.e0 : 0.0;
            const double nxl = nx - e0, nyl = ny - e0, nxh = nx + e0, nyh = ny + e0;
            bool cand = !(nxl > 1.0 && nyl > 1.0);
            if (!cand) {
                const double ax = fabs(sx) + P.e1, ay = fabs(sy) + P.e1;
                // (+ 2 eref: the exact pass reproduces the reference's sums of the RAW values, whose means are only good to
                // ~n eps max|v| -- a phenotype of 1e6 +- 1e-3 makes that visible in t, and such a row must still be offered)
                const double dmax = fabs(sx / nx - sy / ny) + P.e1 * (1.0 / nxl + 1.0 / nyl) + ax * e0 / (nxl * nxl) + ay * e0 / (nyl * nyl) + 2.0 * P.eref;
                const double vx = fmax((qx - P.e2) - ax * ax / nxl, 0.0) / nxh, vy = fmax((qy - P.e2) - ay * ay / nyl, 0.0) / nyh;
                const double sem = vx / (nxh - 1.0) + vy / (nyh - 1.0);
                cand = !(sem > 0.0) || !(dmax / sqrt(sem) <= P.tcrit);
            }
            if (cand) {
                const uint64_t idx = reserve_slot(P);
                P.res_row[idx] = r;
                P.res_nw[idx] = r_nw;
            }
            return;
        }
        const double dx = sx / nx, dy = sy / ny;              // group means minus mu
        const double vx = (qx - sx * dx) / nx, vy = (qy - sy * dy) / ny;  // ddof = 0
        const double sem1 = vx / (nx - 1.0), sem2 = vy / (ny - 1.0);
        const double semsum = sem1 + sem2;
        const double tstat = (dx - dy) / sqrt(semsum);
        // Student's t has heavier tails than the normal, p_t >= erfc(|t|/sqrt 2): rows with
        // |t| <= t_crit (erfc(t_crit/sqrt 2) = cut, solved on the host, less a margin far above the ~1e-15 by which these
        // sums differ from the sample-order ones) cannot pass.  Candidates are stored as (row, n_with) only:
        // ttest_finalize_kernel sums their moments again in the reference's order and decides (keeps erfc /
        // incomplete-beta code, and its ~90 VGPRs, out of this kernel).
        if (!(fabs(tstat) + 2.0 * P.eref / sqrt(semsum) <= P.tcrit)) {   // eref: see ScanArgs; NaN: the exact pass drops it
            const uint64_t idx = reserve_slot(P);
            P.res_row[idx] = r;
            P.res_nw[idx] = r_nw;
        }
    };

    // one copy of process(): the queue is drained after each unrolled batch and, once the rows run out,
    // down to empty (keeps its registers and code out of the streaming part)
    for (uint64_t s0 = wave_global * UNR;; s0 += total_waves * UNR) {
        const bool more = s0 < n_steps;
        while (q >= 64 || (!more && q > 0)) {
            process(q < 64 ? q : 64);
            q = queue_pop64(q_row, q_val, q, lane);
        }
        if (!more) break;
        u32x4 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint64_t row = (s0 + u) * RPW + rsub;
            x[u] = (u32x4)(0u);
            if (HALF) {   // rows `row` and `row + 1` in one 16-byte load
                const u32x4 *pp = reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint2 *>(P.bits) + row);
                if (row + 1 < P.M) x[u] = __builtin_nontemporal_load(pp);
                else if (row < P.M) { const uint2 v = *reinterpret_cast<const uint2 *>(pp); x[u].x = v.x; x[u].y = v.y; }   // the odd last row
            } else if (row < P.M && has_chunk)
                x[u] = (LUT && !PSK_LUT_NT) ? P.bits[row * (uint64_t)P.cpr + g] : __builtin_nontemporal_load(&P.bits[row * (uint64_t)P.cpr + g]);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
          for (int sub = 0; sub < NSUB; sub++) {
            const uint64_t row = (s0 + u) * RPW + rsub + sub;
            const uint64_t xa = sub ? (((uint64_t)x[u].w << 32) | x[u].z) : (((uint64_t)x[u].y << 32) | x[u].x);
            const uint64_t xb = HALF ? 0ull : (((uint64_t)x[u].w << 32) | x[u].z);
            uint32_t cnt = __popcll(xa & mva) + (HALF ? 0u : (uint32_t)__popcll(xb & mvb));
            if (!HALF && P.cpr > GL && row < P.M)
                for (int ch = g + GL; ch < P.cpr; ch += GL) {
                    const u32x4 y = P.bits[row * (uint64_t)P.cpr + ch];
                    cnt += __popcll((((uint64_t)y.y << 32) | y.x) & P.mvalid[2 * ch]) +
                           __popcll((((uint64_t)y.w << 32) | y.z) & P.mvalid[2 * ch + 1]);
                }
#pragma unroll
            for (int d = GL / 2; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
            const int n_w = (int)cnt, n_wo = P.nvalid - (int)cnt;
            const bool freq_ok = (row < P.M) && !(n_w < P.min_samples || n_wo < 2 || n_w > P.max_samples);
            q = queue_rows(freq_ok && g == 0, row, make_int2(n_w, 0), q_row, q_val, q, lane);
        }
    }
}

template <int MODE>
void launch_chi2_mode(int G, dim3 grid, hipStream_t st, const ScanArgs &a)
{
    switch (G) {
    case 0: chi2_scan_kernel<0, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 1: chi2_scan_kernel<1, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 2: chi2_scan_kernel<2, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 4: chi2_scan_kernel<4, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 8: chi2_scan_kernel<8, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 16: chi2_scan_kernel<16, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    case 32: chi2_scan_kernel<32, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    default: chi2_scan_kernel<64, MODE><<<grid, SC_THREADS, 0, st>>>(a); break;
    }
}

// bytes of the nibble table of row_moments_lut for `chunks` 16-byte chunks of a row and NM moments
size_t lut_bytes(int chunks, int nm) { return (size_t)chunks * 32 * 16 * nm * 8; }
// chunks of a row that go through the table: all of them when the table fits the LDS (up to 1536 samples with two
// moments, 1024 with three), else as many as fit -- the rest of the row takes the per-sample form (row_moments_mixed);
// 0 = the per-sample kernels (PSK_NO_LUT, or rows wider than 16 lanes)
int lut_chunks(int cpr, int nm)
{
    if (getenv("PSK_NO_LUT") || cpr > 16) return 0;
    int c = getenv("PSK_LUT_HALF") ? (cpr + 1) / 2 : cpr;
    while (c > 0 && lut_bytes(c, nm) > SC_LUT_MAX_BYTES) c--;
    return c;
}

template <class K>
int launch_lut_kernel(K kern, dim3 grid, size_t lds, hipStream_t st, const ScanArgs &a)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<grid, SC_LUT_THREADS, lds, st>>>(a);
    return 0;
}

void launch_chi2_lut(int G, dim3 grid, hipStream_t st, const ScanArgs &a)
{
    if (a.lut6) {
        const size_t lds6 = lut6_bytes(a.cpr, 2);
        switch (G) {
        case 0: launch_lut_kernel(chi2_scan_kernel<0, 1, true, true>, grid, lds6, st, a); break;
        case 1: launch_lut_kernel(chi2_scan_kernel<1, 1, true, true>, grid, lds6, st, a); break;
        case 2: launch_lut_kernel(chi2_scan_kernel<2, 1, true, true>, grid, lds6, st, a); break;
        case 4: launch_lut_kernel(chi2_scan_kernel<4, 1, true, true>, grid, lds6, st, a); break;
        case 8: launch_lut_kernel(chi2_scan_kernel<8, 1, true, true>, grid, lds6, st, a); break;
        default: launch_lut_kernel(chi2_scan_kernel<16, 1, true, true>, grid, lds6, st, a); break;
        }
        return;
    }
    const size_t lds = lut_bytes(a.c_lut, 2);
    switch (G) {
    case 0: launch_lut_kernel(chi2_scan_kernel<0, 1, true>, grid, lds, st, a); break;
    case 1: launch_lut_kernel(chi2_scan_kernel<1, 1, true>, grid, lds, st, a); break;
    case 2: launch_lut_kernel(chi2_scan_kernel<2, 1, true>, grid, lds, st, a); break;
    case 4: launch_lut_kernel(chi2_scan_kernel<4, 1, true>, grid, lds, st, a); break;
    case 8: launch_lut_kernel(chi2_scan_kernel<8, 1, true>, grid, lds, st, a); break;
    default: launch_lut_kernel(chi2_scan_kernel<16, 1, true>, grid, lds, st, a); break;
    }
}

void launch_chi2(int mode, int G, dim3 grid, hipStream_t st, const ScanArgs &a)
{
    if (mode == 1) {
        if (a.lut || a.lut6) launch_chi2_lut(G, grid, st, a);
        else launch_chi2_mode<1>(G, grid, st, a);
        chi2w_finalize_kernel<<<SC_NSEG, SC_FIN_THREADS, 0, st>>>(a);
    }
    else if (mode == 2) launch_chi2_mode<2>(G, grid, st, a);
    else launch_chi2_mode<0>(G, grid, st, a);
}

// Kernel form of a chi2 scan.  Unit weights: MODE 2 (queued candidates) when many rows are expected to pass the
// pre-test -- the last chi2 scan of this matrix kept more than 0.1 % of the rows, or, with no history, the keep rule
// itself lets that many through under the null hypothesis (p < cut holds for a fraction `cut` of unassociated rows).
// PSK_CHI2_MODE=0|2 forces one (A/B runs).
int pick_chi2_mode(const psk_ctx *ctx, bool weighted, double pcut, double pcut_bonf, int omit_B)
{
    if (weighted) return 1;
    const char *env = getenv("PSK_CHI2_MODE");   // read per call: tests cross the two forms inside one process
    const int forced = env ? atoi(env) : -1;
    if (forced == 0 || forced == 2) return forced;
    if (ctx->dense_hint >= 0) return ctx->dense_hint ? 2 : 0;
    double expect = pcut_bonf;
    if (omit_B && pcut > expect) expect = pcut;
    return expect > 1e-3 ? 2 : 0;
}

// Second pass of the Welch scan: one workgroup per result segment, one candidate per lane.  The candidate's moments are
// summed AGAIN in the reference's order -- conduct_t_test / get_samples_distribution_for_ttest (modeling.py:716-757)
// hand the two groups' values and weights, in sample order, to a weighted DescrStatsW: per group sum w and sum w v, the
// weighted mean, then sum w (v - mean)^2 (ddof = 0), std_meandiff_separatevar and the Satterthwaite df -- with every
// operation an IEEE double operation in that order (this file is compiled with -ffp-contract=off; a weight times 1.0 or
// 0.0 is exact, so `fma(present ? 1 : 0, term, acc)` IS the conditional addition), so t, the two means, round(t, 2)
// and the "%.2E" of the p-value follow from the same bits as a sample-order CPU evaluation (r02: moments from the scan kernel's nibble-table
// sums, ~1e-15 off, and up to two rows flipping at the cut).  Then the two-sided p, the keep rule p < cut / M
// (modeling.py:738) and the compaction of the segment in place.  Candidates are the rows whose scan-kernel |t| exceeds
// a bound no passing row can be under (psk_ttest_scan: t_crit), so the scan kernel's own sums decide nothing.
template <bool WT>
__global__ __launch_bounds__(SC_FIN_THREADS) void ttest_finalize_kernel(const ScanArgs P)
{
    __shared__ uint32_t scan_lds[SC_FIN_THREADS / 64];
    __shared__ uint32_t s_out;
    __shared__ double2 s_tab[SC_FIN_BLK * 128];   // pass 1: {weight, weight * value}, pass 2: {weight, value} of the samples of the current block (broadcast reads)
    const uint32_t seg = blockIdx.x;
    const uint32_t c = P.counter[seg * SC_CNT_STRIDE];
    const uint64_t base = (uint64_t)seg * P.seg_cap;
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    for (uint32_t s0 = 0; s0 < c; s0 += SC_FIN_THREADS) {
        const uint32_t i = s0 + threadIdx.x;
        const bool valid = i < c;
        const uint64_t row = valid ? P.res_row[base + i] : 0;
        const int32_t nw = valid ? P.res_nw[base + i] : 0;
        const bool wave_any = __any(valid);
        const u32x4 *rp = P.half ? sc_row_ptr<true>(P, row) : sc_row_ptr<false>(P, row);
        // One lane walks its candidate's 2 x n_samples dependent additions (rocprof, r03: 82 us at 1,024 samples whatever the
        // number of candidates).  A weight times 1.0 or 0.0 is exact, so fma(present ? 1 : 0, term, acc) IS the conditional
        // addition; the product w v comes out of the staged table, and with unit weights the two weight sums are the counts
        // the scan kernel already has.  Tried and dropped (r03): a branch on the bit instead of the 0/1 factors (94 / 107 us:
        // both sides of a divergent branch issue), and chains of precomputed addends summed by one lane per chain (69 us per
        // two batches of four candidates, and any segment beyond the batches still pays the 82).
        double nx = 0.0, ny = 0.0, sx = 0.0, sy = 0.0, qx = 0.0, qy = 0.0, mx = 0.0, my = 0.0;
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                if (!WT) { nx = (double)nw; ny = (double)(P.nvalid - nw); }   // sums of ones: exact
                mx = sx / nx; my = sy / ny;
            }
            for (int c0 = 0; c0 < P.cpr; c0 += SC_FIN_BLK) {
                const int nc = P.cpr - c0 < SC_FIN_BLK ? P.cpr - c0 : SC_FIN_BLK;
                __syncthreads();   // the previous block has been consumed
                for (int e = threadIdx.x; e < nc * 128; e += SC_FIN_THREADS) {
                    double2 t = reinterpret_cast<const double2 *>(P.raw)[(size_t)c0 * 128 + e];   // NA and padding: {0, 0}
                    if (pass == 0) t.y = t.x * t.y;
                    s_tab[e] = t;
                }
                __syncthreads();
                if (!wave_any) continue;
                // the row's chunks are requested two ahead: read where they are used, every 16-byte chunk cost this lone
                // lane a whole memory latency (16 of them per candidate at 1,024 samples: half of the kernel's 82 us)
                u32x4 y0 = P.half ? sc_ld_chunk<true>(rp, 0) : rp[c0], y1 = nc > 1 ? rp[c0 + 1] : (u32x4)(0u);   // (an 8-byte row: the table's samples 64 ... 127 are {0, 0})
                for (int ch = 0; ch < nc; ch++) {
                    const u32x4 y = y0;
                    y0 = y1;
                    if (ch + 2 < nc) y1 = rp[c0 + ch + 2];
                    const uint32_t w4[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int h = 0; h < 4; h++) {
#pragma unroll 8
                        for (int sb = 0; sb < 32; sb++) {
                            const uint32_t fh = (uint32_t)(((int32_t)(w4[h] << (31 - sb))) >> 31) & 0x3FF00000u;
                            const double f = __hiloint2double((int)fh, 0), g = __hiloint2double((int)(fh ^ 0x3FF00000u), 0);
                            const double2 t = s_tab[ch * 128 + h * 32 + sb];
                            if (pass == 0) {
                                if (WT) { nx = fma(f, t.x, nx); ny = fma(g, t.x, ny); }
                                sx = fma(f, t.y, sx);
                                sy = fma(g, t.y, sy);
                            } else {
                                const double d = t.y - (fh ? mx : my);
                                // unit weights: (1 d) d = d d; an NA sample (weight 0 in the table) adds nothing to either group
                                const double term = WT ? (t.x * d) * d : (t.x != 0.0 ? d * d : 0.0);
                                qx = fma(f, term, qx);
                                qy = fma(g, term, qy);
                            }
                        }
                    }
                }
            }
        }
        double tstat = 0.0, p = 1.0;
        bool keep = false;
        if (valid) {
            const double vx = qx / nx, vy = qy / ny;          // ddof = 0
            const double sem1 = vx / (nx - 1.0), sem2 = vy / (ny - 1.0);
            const double semsum = sem1 + sem2;
            tstat = (mx - my) / sqrt(semsum);
            const double z1 = (sem1 / semsum) * (sem1 / semsum) / (nx - 1.0);
            const double z2 = (sem2 / semsum) * (sem2 / semsum) / (ny - 1.0);
            p = dev_t_two_sided_p(tstat, 1.0 / (z1 + z2));
            keep = p < P.pcut_bonf;
        }
        uint32_t tot;
        const uint32_t pos = psk_block_excl_scan_u32<SC_FIN_THREADS>(keep ? 1u : 0u, &tot, scan_lds);  // barriers inside
        const uint32_t out = s_out;
        if (keep) {
            const uint64_t o = base + out + pos;  // <= base + i: compaction only moves entries down
            P.res_row[o] = row; P.res_stat[o] = tstat; P.res_p[o] = p; P.res_mx[o] = mx; P.res_my[o] = my; P.res_nw[o] = nw;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_out = out + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        P.counter[seg * SC_CNT_STRIDE] = 0;  // re-armed for the next scan
        P.final_counts[seg] = s_out;
        P.host_counts[seg] = s_out;
    }
}

template <bool WT>
void launch_ttest_w(int G, dim3 grid, hipStream_t st, const ScanArgs &a, double mu)
{
    switch (G) {
    case 0: ttest_scan_kernel<0, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 1: ttest_scan_kernel<1, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 2: ttest_scan_kernel<2, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 4: ttest_scan_kernel<4, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 8: ttest_scan_kernel<8, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 16: ttest_scan_kernel<16, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    case 32: ttest_scan_kernel<32, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    default: ttest_scan_kernel<64, WT><<<grid, SC_THREADS, 0, st>>>(a, mu); break;
    }
}

template <class K>
void launch_lut_ttest(K kern, dim3 grid, size_t lds, hipStream_t st, const ScanArgs &a, double mu)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<grid, SC_LUT_THREADS, lds, st>>>(a, mu);
}

template <bool WT>
void launch_ttest_lut(int G, dim3 grid, hipStream_t st, const ScanArgs &a, double mu)
{
    if (a.lut6) {
        const size_t lds6 = lut6_bytes(a.cpr, WT ? 3 : 2);
        switch (G) {
        case 0: launch_lut_ttest(ttest_scan_kernel<0, WT, true, true>, grid, lds6, st, a, mu); break;
        case 1: launch_lut_ttest(ttest_scan_kernel<1, WT, true, true>, grid, lds6, st, a, mu); break;
        case 2: launch_lut_ttest(ttest_scan_kernel<2, WT, true, true>, grid, lds6, st, a, mu); break;
        case 4: launch_lut_ttest(ttest_scan_kernel<4, WT, true, true>, grid, lds6, st, a, mu); break;
        case 8: launch_lut_ttest(ttest_scan_kernel<8, WT, true, true>, grid, lds6, st, a, mu); break;
        default: launch_lut_ttest(ttest_scan_kernel<16, WT, true, true>, grid, lds6, st, a, mu); break;
        }
        return;
    }
    const size_t lds = lut_bytes(a.c_lut, WT ? 3 : 2);
    switch (G) {
    case 0: launch_lut_ttest(ttest_scan_kernel<0, WT, true>, grid, lds, st, a, mu); break;
    case 1: launch_lut_ttest(ttest_scan_kernel<1, WT, true>, grid, lds, st, a, mu); break;
    case 2: launch_lut_ttest(ttest_scan_kernel<2, WT, true>, grid, lds, st, a, mu); break;
    case 4: launch_lut_ttest(ttest_scan_kernel<4, WT, true>, grid, lds, st, a, mu); break;
    case 8: launch_lut_ttest(ttest_scan_kernel<8, WT, true>, grid, lds, st, a, mu); break;
    default: launch_lut_ttest(ttest_scan_kernel<16, WT, true>, grid, lds, st, a, mu); break;
    }
}

void launch_ttest(int G, dim3 grid, hipStream_t st, const ScanArgs &a, double mu, bool weighted)
{
    if (a.lut || a.lut6) {
        if (weighted) launch_ttest_lut<true>(G, grid, st, a, mu);
        else launch_ttest_lut<false>(G, grid, st, a, mu);
    } else if (weighted) launch_ttest_w<true>(G, grid, st, a, mu);
    else launch_ttest_w<false>(G, grid, st, a, mu);
    if (weighted) ttest_finalize_kernel<true><<<SC_NSEG, SC_FIN_THREADS, 0, st>>>(a);
    else ttest_finalize_kernel<false><<<SC_NSEG, SC_FIN_THREADS, 0, st>>>(a);
}

// builds the nibble table of `tab` (cpr * 128 samples x nm moments, already on the device) into ctx->lut
int build_moment_lut(psk_ctx *ctx, const double *tab, int chunks, int nm, const double **lut_out)
{
    const int n_groups = chunks * 32;
    PSK_TRY(dev_reserve(ctx, ctx->lut, lut_bytes(chunks, nm)));
    if (nm == 2) moment_lut_kernel<2><<<div_up((uint64_t)n_groups * 16, 256), 256, 0, ctx->stream>>>(tab, n_groups, ctx->lut.as<double>());
    else moment_lut_kernel<3><<<div_up((uint64_t)n_groups * 16, 256), 256, 0, ctx->stream>>>(tab, n_groups, ctx->lut.as<double>());
    PSK_HIP(ctx, hipGetLastError());
    *lut_out = ctx->lut.as<double>();
    return PSK_OK;
}

// the six-bit f32 table of `tab` for a whole row (cpr chunks), when it fits the LDS beside the kernels' queues; PSK_LUT_F64
// keeps the nibble table in f64 (A/B runs).  *lut6_out = nullptr: not this time.
int build_moment_lut6(psk_ctx *ctx, const double *tab, int cpr, int nm, const float **lut6_out)
{
    *lut6_out = nullptr;
    if (getenv("PSK_NO_LUT") || getenv("PSK_LUT_F64") || cpr > 16 || lut6_bytes(cpr, nm) > SC_LUT_MAX_BYTES) return PSK_OK;
    PSK_TRY(dev_reserve(ctx, ctx->lut, lut6_bytes(cpr, nm)));
    const int n = cpr * SC_L6_ENTRIES;
    if (nm == 2) moment_lut6_kernel<2><<<div_up((uint64_t)n, 256), 256, 0, ctx->stream>>>(tab, cpr, ctx->lut.as<float>());
    else moment_lut6_kernel<3><<<div_up((uint64_t)n, 256), 256, 0, ctx->stream>>>(tab, cpr, ctx->lut.as<float>());
    PSK_HIP(ctx, hipGetLastError());
    *lut6_out = ctx->lut.as<float>();
    return PSK_OK;
}
// (additions per accumulator + rounding of the entry and of the final sums) x 2^-24, with room: what an f32 sum of
// row_moments_f32 may be off by, relative to the sum of the absolute values of ALL the terms of the table
double lut6_gamma(int cpr) { return ((double)(cpr * 22) / 2.0 + 8.0) * 5.9604644775390625e-08 * 1.01; }

// u64 words of a phenotype mask / 64-sample blocks of a per-sample table: the row's words rounded up to a whole chunk
int mask_words(const psk_ctx *ctx) { return (ctx->wpr + 1) & ~1; }

// lanes that own one row; 0 = half a lane (8-byte rows: chi2_scan_kernel's G = 0)
int group_lanes(const ScanArgs &a)
{
    if (a.half) return 0;
    int G = 1;
    while (G < a.cpr && G < 64) G <<= 1;
    return G;
}

// result arrays (SoA) inside ctx->res: row u64 | stat f64 | p f64 | mx f64 | my f64 | nw i32, each
// SC_NSEG * seg_cap entries; seg_cap bounds the rows the blocks of one segment can visit
int setup_results(psk_ctx *ctx, ScanArgs &a, dim3 grid, int G, int unroll, int set, int threads = SC_THREADS)
{
    const uint64_t rpw = sc_rpw(G);
    const uint64_t n_steps = (a.M + rpw - 1) / rpw;
    const uint64_t total_waves = (uint64_t)grid.x * (threads / 64);
    const uint64_t iters = (n_steps + total_waves * unroll - 1) / (total_waves * unroll);
    const uint64_t blocks_per_seg = ((uint64_t)grid.x + SC_NSEG - 1) / SC_NSEG;
    uint64_t seg_cap = blocks_per_seg * (threads / 64) * iters * unroll * rpw;
    if (seg_cap < 64) seg_cap = 64;
    if (seg_cap >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "result segment too large");
    const uint64_t cap = seg_cap * SC_NSEG;
    DevBuf &rb = ctx->slot[set].res;
    PSK_TRY(dev_reserve(ctx, rb, cap * 44 + 64));
    uint8_t *b = rb.as<uint8_t>();
    a.res_row = reinterpret_cast<uint64_t *>(b);
    a.res_stat = reinterpret_cast<double *>(b + cap * 8);
    a.res_p = reinterpret_cast<double *>(b + cap * 16);
    a.res_mx = reinterpret_cast<double *>(b + cap * 24);
    a.res_my = reinterpret_cast<double *>(b + cap * 32);
    a.res_nw = reinterpret_cast<int32_t *>(b + cap * 40);
    if (!ctx->res_count.p) {  // counters re-arm themselves at the end of every scan: zeroed once
        PSK_TRY(dev_reserve(ctx, ctx->res_count, (SC_NSEG * SC_CNT_STRIDE + 2 * SC_NSEG) * 4));
        PSK_HIP(ctx, hipMemsetAsync(ctx->res_count.p, 0, (SC_NSEG * SC_CNT_STRIDE + 2 * SC_NSEG) * 4, ctx->stream));
    }
    if (!ctx->cnt_pinned) PSK_HIP(ctx, hipHostMalloc(&ctx->cnt_pinned, 2 * SC_NSEG * 4, hipHostMallocDefault));
    a.counter = ctx->res_count.as<uint32_t>();
    a.final_counts = a.counter + SC_NSEG * SC_CNT_STRIDE + set * SC_NSEG;  // one compact array per result set
    void *hc = nullptr;
    PSK_HIP(ctx, hipHostGetDevicePointer(&hc, ctx->cnt_pinned, 0));
    a.host_counts = static_cast<uint32_t *>(hc) + set * SC_NSEG;
    a.seg_cap = (uint32_t)seg_cap;
    ctx->slot[set].seg_cap = seg_cap;
    if (set == ctx->res_set) ctx->results_valid = false;  // the last ended scan's results are about to go
    return PSK_OK;
}

// per-segment counts of the scan that wrote result set `set`, as its kernels left them in pinned host memory (after
// that scan has been waited for); n_pass = their sum.  The set becomes the one the result calls read.
int fetch_counts(psk_ctx *ctx, int set)
{
    const uint32_t *raw = static_cast<const uint32_t *>(ctx->cnt_pinned) + set * SC_NSEG;
    const uint64_t seg_cap = ctx->slot[set].seg_cap;
    ctx->seg_counts.assign(SC_NSEG, 0);
    uint64_t tot = 0;
    for (int s = 0; s < SC_NSEG; s++) {
        const uint32_t c = raw[s];
        if (c > seg_cap) return psk_fail(ctx, PSK_ERANGE, "result segment %d overflowed (%u > %llu)", s, c, (unsigned long long)seg_cap);
        ctx->seg_counts[s] = c;
        tot += c;
    }
    ctx->n_pass = tot;
    ctx->res_set = set;
    ctx->res_seg_cap = seg_cap;
    ctx->results_valid = true;
    return PSK_OK;
}

// one block per segment: copies the segment's entries to their place in the contiguous arrays
__global__ void pack_segments_kernel(const uint8_t *__restrict__ src, uint64_t cap, uint32_t seg_cap,
                                     const uint32_t *__restrict__ counts, const uint64_t *__restrict__ offsets,
                                     uint8_t *__restrict__ dst, uint64_t n, const uint64_t *__restrict__ union_words,
                                     uint64_t *__restrict__ words_out)
{
    const uint32_t seg = blockIdx.x;
    const uint32_t c = counts[seg];
    const uint64_t in0 = (uint64_t)seg * seg_cap, out0 = offsets[seg];
    const uint64_t *r = reinterpret_cast<const uint64_t *>(src);
    uint64_t *d = reinterpret_cast<uint64_t *>(dst);
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
#pragma unroll
        for (int f = 0; f < 5; f++) d[(uint64_t)f * n + out0 + i] = r[(uint64_t)f * cap + in0 + i];
        reinterpret_cast<int32_t *>(dst + 40 * n)[out0 + i] = reinterpret_cast<const int32_t *>(src + 40 * cap)[in0 + i];
        words_out[out0 + i] = union_words[r[in0 + i]];  // the k-mer word of the surviving row
    }
}

dim3 scan_grid(const psk_ctx *ctx, uint64_t M, int G, int unroll, bool lut = false)
{
    // the table-in-LDS form: one 1024-thread workgroup per CU (its 64-120 KB of LDS admit no second one), and one
    // per result segment at least
    if (lut) return dim3((unsigned)std::max(SC_NSEG, ctx->n_cu > 0 ? ctx->n_cu : 256));
    const uint64_t rpw = sc_rpw(G);
    const uint64_t steps = (M + rpw - 1) / rpw;
    const uint64_t waves = (steps + unroll - 1) / unroll;
    uint64_t blocks = (waves + SC_THREADS / 64 - 1) / (SC_THREADS / 64);
    static const int mult = [] { const char *e = getenv("PSK_GRID_MULT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : PSK_SC_GRID_MULT; }();
    const uint64_t cap = (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * mult;
    if (blocks > cap) blocks = cap;
    if (blocks < SC_NSEG) blocks = SC_NSEG;  // every result segment needs a workgroup to publish its count
    return dim3((unsigned)blocks);
}

// The result set the next scan writes: the one no scan in flight is writing; with none in flight, the one no
// asynchronous export (psk_export_survivors_async) is still reading and -- keep_results, the two-call form -- not the
// one that holds the last ended scan's results, so that the caller can launch the next scan BEFORE it reads those.
// If the set has an export pending, the scan waits for that export on the device.
int pick_result_set(psk_ctx *ctx, int *set_out, bool keep_results = false)
{
    int set;
    if (ctx->n_in_flight) set = ctx->slot[0].in_flight ? 1 : 0;
    else if (ctx->slot[ctx->res_set].export_pending || (keep_results && ctx->results_valid)) set = ctx->res_set ^ 1;
    else set = ctx->res_set;
    ScanSlot &sl = ctx->slot[set];
    if (sl.export_pending) {
        PSK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.ev_export, 0));
        sl.export_pending = false;
    }
    if (!sl.ev0) {
        PSK_HIP(ctx, hipEventCreate(&sl.ev0));
        PSK_HIP(ctx, hipEventCreate(&sl.ev1));
    }
    *set_out = set;
    return PSK_OK;
}

int run_chi2(psk_ctx *ctx, ScanArgs &a, bool weighted, int reps, double *ms_total, double *ms_each = nullptr)
{
    const int G = group_lanes(a);
    const dim3 grid = scan_grid(ctx, a.M, G, SC_UNROLL, a.lut != nullptr || a.lut6 != nullptr);
    *ms_total = 0;
    for (int r = 0; r < reps; r++) {
        PSK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        launch_chi2(pick_chi2_mode(ctx, weighted, a.pcut, a.pcut_bonf, a.omit_B), G, grid, ctx->stream, a);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the kernel has written the counts to pinned memory
        float ms = 0;
        PSK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *ms_total += ms;
        if (ms_each) ms_each[r] = ms;
    }
    return PSK_OK;
}

}  // namespace

static int fill_chi2_args(psk_ctx *ctx, ScanArgs &a, int set)
{
    const ScanParams &L = ctx->last;
    a = ScanArgs();
    a.bits = reinterpret_cast<const u32x4 *>(ctx->bits.p);
    a.M = ctx->n_kmers;
    const int mw = mask_words(ctx);
    a.cpr = mw / 2;
    a.half = ctx->wpr == 1;
    a.m1 = ctx->mask1.as<uint64_t>();
    a.m0 = a.m1 + mw;
    a.tab = reinterpret_cast<const double *>(a.m1 + 2 * (size_t)mw);  // [sample][w if pheno 1 | w if pheno 0]
    a.inline_masks = L.inline_masks;
    if (L.inline_masks) { memcpy(a.m1_inl, L.m1, sizeof(a.m1_inl)); memcpy(a.m0_inl, L.m0, sizeof(a.m0_inl)); }
    a.min_samples = L.min_samples;
    a.max_samples = L.max_samples;
    a.pcut = L.pvalue_cutoff;
    a.pcut_bonf = L.pvalue_cutoff / (double)L.n_kmers_global;
    a.omit_B = L.omit_B;
    double pmax = a.pcut_bonf;
    if (L.omit_B && a.pcut > pmax) pmax = a.pcut;
    if (pmax >= 1.0) a.thr = 0.0;
    else if (pmax <= 0.0) a.thr = INFINITY;
    else a.thr = -2.0 * log(pmax);
    const int G = group_lanes(a);
    a.lut6 = (L.weighted && ctx->lut6_valid) ? ctx->lut.as<float>() : nullptr;
    a.lut = (L.weighted && ctx->lut_valid && !a.lut6) ? ctx->lut.as<double>() : nullptr;
    a.c_lut = a.lut ? lut_chunks(a.cpr, 2) : 0;
    a.W1 = L.W1; a.W0 = L.W0;
    if (a.lut6) {   // what the f32 class-weight sums may be off by (row_moments_f32)
        a.e0 = lut6_gamma(a.cpr) * L.W1 + 1e-36;
        a.e1 = lut6_gamma(a.cpr) * L.W0 + 1e-36;
    }
    const bool table = a.lut != nullptr || a.lut6 != nullptr;
    return setup_results(ctx, a, scan_grid(ctx, a.M, G, SC_UNROLL, table), G, table ? lut_unroll(G) : SC_UNROLL, set,
                         table ? SC_LUT_THREADS : SC_THREADS);
}

// Launches the scan and returns without waiting; psk_scan_end collects it.  Lets a caller queue other work (the
// survivor exchange of the previous scan) while the kernel streams the matrix.
static int chi2_scan_launch(psk_ctx *ctx, const int8_t *pheno, const double *weights, int min_samples, int max_samples,
                            double pvalue_cutoff, int omit_B, uint64_t n_kmers_global, bool keep_results)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight >= 2) return psk_fail(ctx, PSK_ESTATE, "two scans are in flight (psk_scan_end first)");
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix (psk_build_presence first)");
    if (!pheno) return psk_fail(ctx, PSK_EINVAL, "null phenotype vector");
    if (n_kmers_global == 0) n_kmers_global = ctx->n_kmers ? ctx->n_kmers : 1;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const int N = ctx->n_samples, wpr = mask_words(ctx);   // masks and tables: whole 16-byte chunks, also for 8-byte rows
    // one pinned staging block [m1 | m0 | w1 | w0] and ONE stream-ordered upload (weights only when given)
    const size_t n_mask = 2 * (size_t)wpr, n_w = 2 * (size_t)wpr * 64;
    const size_t stage_bytes = (n_mask + n_w) * 8;
    int set = 0;
    PSK_TRY(pick_result_set(ctx, &set, keep_results));
    if (2 * stage_bytes > ctx->scan_pinned_cap) {  // one staging block per result set: an upload may still be queued
        if (ctx->n_in_flight) PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scan_pinned) (void)hipHostFree(ctx->scan_pinned);
        ctx->scan_pinned = nullptr;
        ctx->scan_pinned_cap = 0;
        PSK_HIP(ctx, hipHostMalloc(&ctx->scan_pinned, 2 * stage_bytes, hipHostMallocDefault));
        ctx->scan_pinned_cap = 2 * stage_bytes;
    }
    uint64_t *m1 = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(ctx->scan_pinned) + set * stage_bytes), *m0 = m1 + wpr;
    double *w = reinterpret_cast<double *>(m1 + n_mask);
    memset(m1, 0, weights ? stage_bytes : n_mask * 8);
    double W1 = 0, W0 = 0;
    int n1 = 0, n0 = 0;
    for (int i = 0; i < N; i++) {
        const double wi = weights ? weights[i] : 1.0;
        if (pheno[i] == 1) { m1[i >> 6] |= 1ull << (i & 63); if (weights) w[2 * (size_t)i] = wi; W1 += wi; n1++; }
        else if (pheno[i] == 0) { m0[i >> 6] |= 1ull << (i & 63); if (weights) w[2 * (size_t)i + 1] = wi; W0 += wi; n0++; }
    }
    // up to 1024 samples: the masks ride in the kernel arguments and an unweighted scan uploads nothing
    ctx->last.inline_masks = wpr <= SC_INL_WORDS ? 1 : 0;
    if (ctx->last.inline_masks) {
        memset(ctx->last.m1, 0, sizeof(ctx->last.m1));
        memset(ctx->last.m0, 0, sizeof(ctx->last.m0));
        memcpy(ctx->last.m1, m1, (size_t)wpr * 8);
        memcpy(ctx->last.m0, m0, (size_t)wpr * 8);
    }
    PSK_TRY(dev_reserve(ctx, ctx->mask1, stage_bytes));
    if (weights || !ctx->last.inline_masks)
        PSK_HIP(ctx, hipMemcpyAsync(ctx->mask1.p, m1, weights ? stage_bytes : n_mask * 8, hipMemcpyHostToDevice, ctx->stream));

    ctx->lut_valid = false;
    ctx->lut6_valid = false;
    if (weights) {   // class-weight sums from a table in LDS: six-bit groups in f32 (row_moments_f32) when a row's table fits, else nibbles in f64
        const double *wtab = reinterpret_cast<const double *>(ctx->mask1.as<uint64_t>() + 2 * (size_t)wpr);
        const float *lut6 = nullptr;
        PSK_TRY(build_moment_lut6(ctx, wtab, wpr / 2, 2, &lut6));
        if (lut6) ctx->lut6_valid = true;
        else if (lut_chunks(wpr / 2, 2) > 0) {
            const double *lut = nullptr;
            PSK_TRY(build_moment_lut(ctx, wtab, lut_chunks(wpr / 2, 2), 2, &lut));
            ctx->lut_valid = true;
        }
    }
    ctx->last.valid = true;
    ctx->last.weighted = weights != nullptr;
    ctx->last.min_samples = min_samples;
    ctx->last.max_samples = max_samples;
    ctx->last.pvalue_cutoff = pvalue_cutoff;
    ctx->last.omit_B = omit_B ? 1 : 0;
    ctx->last.n_kmers_global = n_kmers_global;
    ctx->last.n1 = n1; ctx->last.n0 = n0; ctx->last.W1 = W1; ctx->last.W0 = W0;
    ScanArgs a;
    PSK_TRY(fill_chi2_args(ctx, a, set));
    a.n1 = n1; a.n0 = n0;
    ctx->last_scan_kind = 1;
    if (ctx->n_kmers) {
        ScanSlot &sl = ctx->slot[set];
        const int G = group_lanes(a);
        const dim3 grid = scan_grid(ctx, a.M, G, SC_UNROLL, a.lut != nullptr || a.lut6 != nullptr);
        PSK_HIP(ctx, hipEventRecord(sl.ev0, ctx->stream));
        launch_chi2(pick_chi2_mode(ctx, ctx->last.weighted, a.pcut, a.pcut_bonf, a.omit_B), G, grid, ctx->stream, a);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(sl.ev1, ctx->stream));
        sl.in_flight = true;
        sl.seq = ++ctx->scan_seq;
        ctx->n_in_flight++;
    } else {  // nothing to scan: an empty result, at once
        ctx->n_pass = 0;
        ctx->seg_counts.assign(SC_NSEG, 0);
        ctx->res_set = set;
        ctx->results_valid = true;
    }
    return PSK_OK;
}

extern "C" int psk_chi2_scan_begin(psk_ctx *ctx, const int8_t *pheno, const double *weights, int min_samples,
                                   int max_samples, double pvalue_cutoff, int omit_B, uint64_t n_kmers_global)
{
    return chi2_scan_launch(ctx, pheno, weights, min_samples, max_samples, pvalue_cutoff, omit_B, n_kmers_global, true);
}

extern "C" int psk_scan_end(psk_ctx *ctx, uint64_t *n_pass)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight) {  // the oldest scan in flight
        PSK_HIP(ctx, hipSetDevice(ctx->device));
        int set = ctx->slot[0].in_flight ? 0 : 1;
        if (ctx->slot[0].in_flight && ctx->slot[1].in_flight && ctx->slot[1].seq < ctx->slot[0].seq) set = 1;
        ScanSlot &sl = ctx->slot[set];
        PSK_HIP(ctx, hipEventSynchronize(sl.ev1));  // its kernels have written the counts to pinned memory
        sl.in_flight = false;
        ctx->n_in_flight--;
        float ms = 0;
        PSK_HIP(ctx, hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
        ctx->last_scan_ms = ms;
        PSK_TRY(fetch_counts(ctx, set));
        ctx->dense_hint = ctx->n_pass * 1000 > ctx->n_kmers ? 1 : 0;  // only chi2 scans come through here
    }
    if (n_pass) *n_pass = ctx->n_pass;
    return PSK_OK;
}

extern "C" int psk_chi2_scan(psk_ctx *ctx, const int8_t *pheno, const double *weights, int min_samples,
                             int max_samples, double pvalue_cutoff, int omit_B, uint64_t n_kmers_global,
                             uint64_t *n_pass)
{
    if (ctx && ctx->n_in_flight) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight (psk_scan_end first)");
    PSK_TRY(chi2_scan_launch(ctx, pheno, weights, min_samples, max_samples, pvalue_cutoff, omit_B, n_kmers_global, false));
    return psk_scan_end(ctx, n_pass);
}


static int rescan(psk_ctx *ctx, int reps, double *mean_ms, double *ms_each)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight (psk_scan_end first)");
    if (!ctx->have_presence || !ctx->last.valid || ctx->last_scan_kind != 1)
        return psk_fail(ctx, PSK_ESTATE, "no chi2 scan to repeat");
    if (reps < 1) return psk_fail(ctx, PSK_EINVAL, "reps must be >= 1");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    int set = 0;
    PSK_TRY(pick_result_set(ctx, &set));
    ScanArgs a;
    PSK_TRY(fill_chi2_args(ctx, a, set));
    a.n1 = ctx->last.n1; a.n0 = ctx->last.n0; a.W1 = ctx->last.W1; a.W0 = ctx->last.W0;
    double ms = 0;
    PSK_TRY(run_chi2(ctx, a, ctx->last.weighted, reps, &ms, ms_each));
    ctx->last_scan_ms = ms / reps;
    PSK_TRY(fetch_counts(ctx, set));
    if (mean_ms) *mean_ms = ms / reps;
    return PSK_OK;
}

extern "C" int psk_rescan_timed(psk_ctx *ctx, int reps, double *mean_ms) { return rescan(ctx, reps, mean_ms, nullptr); }

extern "C" int psk_rescan_times(psk_ctx *ctx, int reps, double *ms_each)
{
    if (ctx && !ms_each) return psk_fail(ctx, PSK_EINVAL, "null output array");
    return rescan(ctx, reps, nullptr, ms_each);
}

extern "C" int psk_ttest_scan(psk_ctx *ctx, const double *pheno, const uint8_t *valid, const double *weights,
                              int min_samples, int max_samples, double pvalue_cutoff, uint64_t n_kmers_global,
                              uint64_t *n_pass)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight (psk_scan_end first)");
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix (psk_build_presence first)");
    if (!pheno || !valid) return psk_fail(ctx, PSK_EINVAL, "null phenotype vector");
    if (n_kmers_global == 0) n_kmers_global = ctx->n_kmers ? ctx->n_kmers : 1;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const int N = ctx->n_samples, wpr = mask_words(ctx);   // masks and tables: whole 16-byte chunks, also for 8-byte rows
    std::vector<uint64_t> mv(wpr, 0);
    bool unit_w = true;
    for (int i = 0; weights && i < N; i++) if (valid[i] && weights[i] != 1.0) unit_w = false;
    const int NM = unit_w ? 2 : 3;
    const size_t raw_off = (size_t)NM * wpr * 64;             // the exact pass's {weight, value} pairs follow the moment table
    std::vector<double> vw(raw_off + (size_t)2 * wpr * 64, 0.0);  // row_moments table: {u, u^2} or {w, w u, w u^2} per sample
    int nvalid = 0;
    double sw = 0.0, swv = 0.0;
    for (int i = 0; i < N; i++) {
        if (!valid[i]) continue;
        const double wi = weights ? weights[i] : 1.0;
        sw += wi;
        swv += wi * pheno[i];
    }
    const double mu = sw > 0 ? swv / sw : 0.0;  // the kernel accumulates moments of (value - mu)
    // ... scaled by a power of two (exact; t does not change) so that the spread is ~1: the f32 table of the candidate
    // pass then neither underflows nor overflows whatever unit the phenotype is in
    double scale = 1.0;
    {
        double ss = 0.0;
        for (int i = 0; i < N; i++) if (valid[i]) { const double u = pheno[i] - mu; ss += (weights ? weights[i] : 1.0) * u * u; }
        const double sd = sw > 0 ? std::sqrt(ss / sw) : 0.0;
        if (sd > 0 && std::isfinite(sd)) { int ex = 0; (void)std::frexp(sd, &ex); scale = std::ldexp(1.0, 1 - ex); }
    }
    double tot_w = 0.0, tot_wu = 0.0, tot_wuu = 0.0, abs_wu = 0.0, max_abs_v = 0.0;
    for (int i = 0; i < N; i++) {
        if (!valid[i]) continue;
        mv[i >> 6] |= 1ull << (i & 63);
        const double u = (pheno[i] - mu) * scale, wi = weights ? weights[i] : 1.0;
        abs_wu += std::fabs(wi * u);
        max_abs_v = std::max(max_abs_v, std::fabs(pheno[i]));
        if (unit_w) { vw[2 * (size_t)i] = u; vw[2 * (size_t)i + 1] = u * u; }
        else { vw[3 * (size_t)i] = wi; vw[3 * (size_t)i + 1] = wi * u; vw[3 * (size_t)i + 2] = wi * u * u; }
        tot_w += wi; tot_wu += wi * u; tot_wuu += wi * u * u;
        vw[raw_off + 2 * (size_t)i] = wi; vw[raw_off + 2 * (size_t)i + 1] = pheno[i];
        nvalid++;
    }
    PSK_TRY(dev_reserve(ctx, ctx->mask1, wpr * 8));
    PSK_TRY(dev_reserve(ctx, ctx->phe, vw.size() * 8));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->mask1.p, mv.data(), wpr * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->phe.p, vw.data(), vw.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ScanArgs a = ScanArgs();
    a.bits = reinterpret_cast<const u32x4 *>(ctx->bits.p);
    a.M = ctx->n_kmers;
    a.cpr = wpr / 2;
    a.half = ctx->wpr == 1;
    a.mvalid = ctx->mask1.as<uint64_t>();
    a.tab = ctx->phe.as<double>();
    a.raw = a.tab + raw_off;
    a.nvalid = nvalid;
    a.min_samples = min_samples;
    a.max_samples = max_samples;
    a.pcut = pvalue_cutoff;
    a.pcut_bonf = pvalue_cutoff / (double)n_kmers_global;
    a.W1 = tot_w; a.W0 = tot_wu; a.thr = tot_wuu;  // totals over the non-NA samples (shifted values)
    {   // t_crit: erfc(t_crit / sqrt 2) = cut / M by bisection (0 when everything may pass)
        const double cut = a.pcut_bonf;
        double lo = 0.0, hi = 40.0;
        if (!(cut < 1.0)) hi = 0.0;
        else if (std::erfc(hi * 0.70710678118654752440) >= cut) lo = hi;  // cut below double's erfc range
        else
            for (int it = 0; it < 200; it++) {
                const double mid = 0.5 * (lo + hi);
                if (std::erfc(mid * 0.70710678118654752440) >= cut) lo = mid; else hi = mid;
            }
        a.tcrit = cut < 1.0 ? lo * (1.0 - 1e-9) : -1.0;  // err on the side of keeping candidates (the exact pass decides)
    }
    int set = 0;
    PSK_TRY(pick_result_set(ctx, &set));
    const int G = group_lanes(a);
    ctx->lut_valid = false;   // the table buffer is shared with the weighted chi2 scan
    ctx->lut6_valid = false;
    a.eref = 4.0 * (double)N * 1.1102230246251565e-16 * max_abs_v * scale;   // in the kernel's (shifted, scaled) units
    PSK_TRY(build_moment_lut6(ctx, a.tab, a.cpr, NM, &a.lut6));
    if (a.lut6) {   // what the f32 sums of the candidate pass may be off by (row_moments_f32)
        const double gm = lut6_gamma(a.cpr);
        a.e0 = gm * tot_w + 1e-36; a.e1 = gm * abs_wu + 1e-36; a.e2 = gm * tot_wuu + 1e-36;
    } else {
        a.c_lut = lut_chunks(a.cpr, NM);
        if (a.c_lut > 0) PSK_TRY(build_moment_lut(ctx, a.tab, a.c_lut, NM, &a.lut));
    }
    const bool table = a.lut != nullptr || a.lut6 != nullptr;
    const dim3 grid = scan_grid(ctx, a.M, G, SC_UNROLL, table);
    PSK_TRY(setup_results(ctx, a, grid, G, table ? lut_unroll(G) : SC_UNROLL, set, table ? SC_LUT_THREADS : SC_THREADS));
    ctx->n_pass = 0;
    ctx->last_scan_kind = 2;
    ctx->last.valid = false;
    if (ctx->n_kmers) {
        PSK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        launch_ttest(G, grid, ctx->stream, a, mu, !unit_w);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        PSK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->last_scan_ms = ms;
        PSK_TRY(fetch_counts(ctx, set));
    } else {
        ctx->seg_counts.assign(SC_NSEG, 0);
        ctx->res_set = set;
        ctx->results_valid = true;
    }
    if (n_pass) *n_pass = ctx->n_pass;
    return PSK_OK;
}

extern "C" int psk_get_results(psk_ctx *ctx, uint64_t *row_idx, uint64_t *words, double *stat, double *p,
                               double *mean_x, double *mean_y, int32_t *n_with, uint64_t cap)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->last_scan_kind) return psk_fail(ctx, PSK_ESTATE, "no scan has been run");
    if (!ctx->results_valid) return psk_fail(ctx, PSK_ESTATE, "no ended scan whose results are still held (psk_scan_end first)");
    const uint64_t n = ctx->n_pass;
    if (cap < n) return psk_fail(ctx, PSK_ERANGE, "buffer too small: %llu < %llu", (unsigned long long)cap,
                                 (unsigned long long)n);
    if (n == 0) return PSK_OK;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    // pack the segments into contiguous SoA arrays of n entries
    {
        std::vector<uint64_t> offs(SC_NSEG);
        uint64_t acc = 0;
        for (int sgm = 0; sgm < SC_NSEG; sgm++) { offs[sgm] = acc; acc += ctx->seg_counts[sgm]; }
        PSK_TRY(dev_reserve(ctx, ctx->res_sorted, n * 52 + 128 + SC_NSEG * 12));
        uint8_t *aux = ctx->res_sorted.as<uint8_t>() + ((n * 52 + 63) & ~63ull);
        uint32_t *d_cnt = reinterpret_cast<uint32_t *>(aux + SC_NSEG * 8);
        PSK_HIP(ctx, hipMemcpyAsync(aux, offs.data(), SC_NSEG * 8, hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(d_cnt, ctx->seg_counts.data(), SC_NSEG * 4, hipMemcpyHostToDevice, ctx->stream));
        pack_segments_kernel<<<SC_NSEG, 256, 0, ctx->stream>>>(ctx->slot[ctx->res_set].res.as<uint8_t>(), ctx->res_seg_cap * SC_NSEG,
                                                             (uint32_t)ctx->res_seg_cap, d_cnt,
                                                             reinterpret_cast<const uint64_t *>(aux),
                                                             ctx->res_sorted.as<uint8_t>(), n,
                                                             ctx->union_words.as<uint64_t>(),
                                                             reinterpret_cast<uint64_t *>(ctx->res_sorted.as<uint8_t>() + ((n * 44 + 7) & ~7ull)));
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const uint64_t c = n;
    const uint8_t *b = ctx->res_sorted.as<uint8_t>();
    std::vector<uint64_t> rows(n);
    std::vector<double> st(n), pv(n), mx(n), my(n);
    std::vector<int32_t> nw(n);
    PSK_HIP(ctx, hipMemcpy(rows.data(), b, n * 8, hipMemcpyDeviceToHost));
    PSK_HIP(ctx, hipMemcpy(st.data(), b + c * 8, n * 8, hipMemcpyDeviceToHost));
    PSK_HIP(ctx, hipMemcpy(pv.data(), b + c * 16, n * 8, hipMemcpyDeviceToHost));
    if (ctx->last_scan_kind == 2) {
        PSK_HIP(ctx, hipMemcpy(mx.data(), b + c * 24, n * 8, hipMemcpyDeviceToHost));
        PSK_HIP(ctx, hipMemcpy(my.data(), b + c * 32, n * 8, hipMemcpyDeviceToHost));
    }
    PSK_HIP(ctx, hipMemcpy(nw.data(), b + c * 40, n * 4, hipMemcpyDeviceToHost));
    // the append order of the scan is not deterministic: order by row (= ascending k-mer)
    std::vector<uint64_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](uint64_t x, uint64_t y) { return rows[x] < rows[y]; });
    std::vector<uint64_t> wbuf;
    if (words) {  // gathered on the device by pack_segments_kernel, same (segment) order as the other columns
        wbuf.resize(n);
        PSK_HIP(ctx, hipMemcpy(wbuf.data(), b + ((n * 44 + 7) & ~7ull), n * 8, hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t j = ord[i];
        if (row_idx) row_idx[i] = rows[j];
        if (words) words[i] = wbuf[j];
        if (stat) stat[i] = st[j];
        if (p) p[i] = pv[j];
        if (mean_x) mean_x[i] = (ctx->last_scan_kind == 2) ? mx[j] : 0.0;
        if (mean_y) mean_y[i] = (ctx->last_scan_kind == 2) ? my[j] : 0.0;
        if (n_with) n_with[i] = nw[j];
    }
    return PSK_OK;
}

// one block per result segment: writes the segment's survivors as AoS records into a caller buffer.
// Segment offsets are computed on the device from the scan's own counters (no host round trip).
__global__ void export_records_kernel(const uint8_t *__restrict__ res, uint64_t cap, uint32_t seg_cap,
                                      const uint32_t *__restrict__ counters, const uint64_t *__restrict__ union_words,
                                      const uint64_t *__restrict__ bits, int wpr, uint64_t *__restrict__ dst,
                                      uint64_t cap_records)
{
    __shared__ uint32_t cnt[SC_NSEG];
    __shared__ uint64_t s_off, s_total;
    const uint32_t seg = blockIdx.x;
    cnt[threadIdx.x] = counters[threadIdx.x];  // the compact final counts; blockDim.x == SC_NSEG
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t off = 0, tot = 0;
        for (int j = 0; j < SC_NSEG; j++) { if (j == (int)seg) off = tot; tot += cnt[j]; }
        s_off = off;
        s_total = tot;
    }
    __syncthreads();
    const uint32_t c = cnt[seg];
    const uint64_t in0 = (uint64_t)seg * seg_cap, out0 = s_off;
    const uint64_t rec_words = 6 + (uint64_t)wpr;
    const uint64_t *f64s = reinterpret_cast<const uint64_t *>(res);  // row / stat / p / mx / my as raw 64-bit patterns
    const int32_t *nw = reinterpret_cast<const int32_t *>(res + 40 * cap);
    if (seg == 0 && threadIdx.x == 0) {  // header record: number of records that follow
        dst[0] = s_total;
        for (uint64_t j = 1; j < rec_words; j++) dst[j] = 0;
    }
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        const uint64_t o = out0 + i;
        if (o >= cap_records) continue;
        uint64_t *rec = dst + (o + 1) * rec_words;
        const uint64_t r = f64s[in0 + i];
        rec[0] = union_words[r];
        rec[1] = f64s[1 * cap + in0 + i];
        rec[2] = f64s[2 * cap + in0 + i];
        rec[3] = f64s[3 * cap + in0 + i];
        rec[4] = f64s[4 * cap + in0 + i];
        rec[5] = (uint64_t)(int64_t)nw[in0 + i];
        for (int w = 0; w < wpr; w++) rec[6 + w] = bits[r * (uint64_t)wpr + w];
    }
}

extern "C" int psk_export_survivors(psk_ctx *ctx, void *device_dst, uint64_t cap_records, uint64_t *n_records)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->last_scan_kind) return psk_fail(ctx, PSK_ESTATE, "no scan has been run");
    if (!ctx->results_valid) return psk_fail(ctx, PSK_ESTATE, "no ended scan whose results are still held (psk_scan_end first)");
    if (!device_dst || cap_records < 1) return psk_fail(ctx, PSK_EINVAL, "bad destination");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    if (n_records) *n_records = ctx->n_pass;
    export_records_kernel<<<SC_NSEG, SC_NSEG, 0, ctx->stream>>>(
        ctx->slot[ctx->res_set].res.as<uint8_t>(), ctx->res_seg_cap * SC_NSEG, (uint32_t)ctx->res_seg_cap,
        ctx->res_count.as<uint32_t>() + SC_NSEG * SC_CNT_STRIDE + ctx->res_set * SC_NSEG, ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>(),
        ctx->wpr, static_cast<uint64_t *>(device_dst),
        cap_records);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

// The same export, queued on the CALLER's stream and not waited for: a collective queued on that stream next (RCCL
// all_gather_into_tensor) is ordered after it without a host synchronisation, and the next scan of this context
// waits (on the device) for the export before it overwrites the result arrays.
extern "C" int psk_export_survivors_async(psk_ctx *ctx, void *device_dst, uint64_t cap_records, void *stream)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->last_scan_kind) return psk_fail(ctx, PSK_ESTATE, "no scan has been run");
    if (!ctx->results_valid) return psk_fail(ctx, PSK_ESTATE, "no ended scan whose results are still held (psk_scan_end first)");
    if (!device_dst || cap_records < 1) return psk_fail(ctx, PSK_EINVAL, "bad destination");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t &ev = ctx->slot[ctx->res_set].ev_export;
    if (!ev) PSK_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    export_records_kernel<<<SC_NSEG, SC_NSEG, 0, st>>>(
        ctx->slot[ctx->res_set].res.as<uint8_t>(), ctx->res_seg_cap * SC_NSEG, (uint32_t)ctx->res_seg_cap,
        ctx->res_count.as<uint32_t>() + SC_NSEG * SC_CNT_STRIDE + ctx->res_set * SC_NSEG, ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>(),
        ctx->wpr, static_cast<uint64_t *>(device_dst), cap_records);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipEventRecord(ev, st));
    ctx->slot[ctx->res_set].export_pending = true;
    return PSK_OK;
}

extern "C" double psk_last_scan_ms(const psk_ctx *ctx) { return ctx ? ctx->last_scan_ms : 0.0; }
