// a1 for small word spaces (2k <= 26, i.e. the reference's default k = 13): direct-address counting instead of a
// sort (replaces bin/glistmaker; modeling.py:303-315).  The 2k-bit word space is cut into buckets of 2^15 values;
// a bucket's counters fit the LDS of one workgroup, so a sample is counted by
//   dc_hist_kernel       tile of 16,384 bases per workgroup: canonical words, slab filter, bucket histogram in LDS;
//                        one returning atomic per non-empty (tile, bucket) reserves the tile's range in the bucket
//   dc_partition_kernel  the same tile again: every word gets its rank inside (tile, bucket) from a returning LDS
//                        atomic, the tile's words are bucketed in LDS and spilled in bucket runs as 15-bit values
//   dc_count_kernel      one workgroup per bucket: 2^15 counters in 64 KB of LDS (two 16-bit counters per word; a
//                        bucket with 65,536 keys or more takes 32-bit counters in two halves), then straight from
//                        the table: the bucket's 4 KB slice of the sample's presence bitmap, the number of distinct
//                        words, and the (word, count) entries with count >= 2
//   dc_totals_kernel     sums over the buckets + offsets of the multi-count entries
//   dc_compact_kernel    (one sample later, once the host has sized the arena block) packs the multi-count entries
// The output is the DENSE list form of psk_internal.h: ascending order is implicit, no radix pass, no run-length
// pass, 12 B per base of key traffic become 2 B, and the list shrinks from 12 B per distinct word to 1 bit per word
// of the space + 8 B per repeated word.  Integer work, HBM / LDS-atomic bound: no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"
#include "kmer_windows.h"

namespace {

constexpr int DT_THREADS = 512;
constexpr int DT_SEG = KW_SEG;                   // window ends per thread (kmer_windows.h)
constexpr int DT_TILE = DT_THREADS * DT_SEG;     // 16,384 bases per workgroup
constexpr int DC_VALS = 1 << DC_VB;
constexpr uint32_t DC_SLOTS = 32;                // pre-zeroed counter slots per buffer set
constexpr size_t DC_DENSE_BYTES_MAX = 64ull << 30;
constexpr size_t DP_LDS_BYTES = (size_t)(DT_TILE + DC_MAX_NB + DC_MAX_NB / 2 + 16) * 4;   // 76 KB: two workgroups per CU
constexpr size_t DCNT_LDS_BYTES = (size_t)(DC_VALS / 2 + 16) * 4;                         // 64 KB + scan scratch

// exclusive scan over a workgroup of DT_THREADS threads, one value per thread
__device__ __forceinline__ uint32_t block_scan512(uint32_t v, uint32_t *total, uint32_t *lds)
{
    return psk_block_excl_scan_u32<DT_THREADS>(v, total, lds);
}

// ---- a group of samples per launch (dense_group_enqueue) ---------------------------------------------------------------
// One 5-Mbp genome is 306 tiles: barely more than one workgroup per CU, so each of its launches lasts as long as ONE tile
// (15 to 21 us) plus its ramp, whatever the GPU could do beside it, and the two single-workgroup kernels are pure launch
// floor -- 62 us of kernels per genome at 1 TB/s (r02).  The kernels below are the same bodies with a sample dimension:
// the tile kernels find their sample from the first tile of each (at most DC_GROUP entries in the kernel arguments), the
// per-bucket kernels take it from blockIdx.y.
constexpr int DC_GROUP = 8;
struct DcItem {
    const uint8_t *clean;
    uint64_t len;
    uint32_t *cnt, *wgoff, *base;
    uint16_t *part;
    uint64_t *bitmap;
    uint32_t *mt_w, *mt_f, *uniq, *multi, *need, *totals, *host_totals, *dst_w, *dst_f;
    uint32_t tile0;   // first tile of the sample in the group's grid
};
struct DcBatch {
    uint32_t n, lo, hi, b0, nb;
    DcItem it[DC_GROUP];
};
__device__ __forceinline__ uint32_t dc_batch_sample(const DcBatch &p, uint32_t tile)
{
    uint32_t s = 0;
#pragma unroll
    for (int q = 1; q < DC_GROUP; q++) s += (q < (int)p.n && tile >= p.it[q].tile0) ? 1u : 0u;
    return s;
}

template <int K>
__device__ __forceinline__ void dc_hist_body(const uint8_t *__restrict__ clean, uint64_t len, uint32_t lo, uint32_t hi, uint32_t b0,
                                             uint32_t nb, uint32_t *__restrict__ cnt, uint32_t *__restrict__ wgoff, const uint32_t tile)
{
    __shared__ uint32_t h[DC_MAX_NB];
    for (uint32_t d = threadIdx.x; d < nb; d += DT_THREADS) h[d] = 0;
    __syncthreads();
    const uint64_t s = ((uint64_t)tile * DT_THREADS + threadIdx.x) * DT_SEG;
    Streams st;
    load_streams(st, clean, len, s);
    ForEachWindow<K, 0>::run(st, lo, hi, [&](int, bool ok, uint32_t w) {
        if (ok) atomicAdd(&h[(w >> DC_VB) - b0], 1u);
    });
    __syncthreads();
    // the (up to four) returning atomics of a thread are issued back to back: one round trip, not four
    uint32_t c[DC_MAX_NB / DT_THREADS], o[DC_MAX_NB / DT_THREADS];
#pragma unroll
    for (int e = 0; e < (int)(DC_MAX_NB / DT_THREADS); e++) {
        const uint32_t d = e * DT_THREADS + threadIdx.x;
        c[e] = d < nb ? h[d] : 0u;
    }
#pragma unroll
    for (int e = 0; e < (int)(DC_MAX_NB / DT_THREADS); e++)
        if (c[e]) o[e] = atomicAdd(&cnt[e * DT_THREADS + threadIdx.x], c[e]);
#pragma unroll
    for (int e = 0; e < (int)(DC_MAX_NB / DT_THREADS); e++)
        if (c[e]) wgoff[(uint64_t)tile * nb + e * DT_THREADS + threadIdx.x] = o[e];
}
template <int K>
__global__ __launch_bounds__(DT_THREADS) void dc_hist_kernel(const uint8_t *__restrict__ clean, uint64_t len, uint32_t lo,
                                                              uint32_t hi, uint32_t b0, uint32_t nb, uint32_t *__restrict__ cnt,
                                                              uint32_t *__restrict__ wgoff)
{
    dc_hist_body<K>(clean, len, lo, hi, b0, nb, cnt, wgoff, blockIdx.x);
}
template <int K>
__global__ __launch_bounds__(DT_THREADS) void dc_hist_batch_kernel(const DcBatch p)
{
    const uint32_t s = dc_batch_sample(p, blockIdx.x);
    const DcItem &it = p.it[s];
    dc_hist_body<K>(it.clean, it.len, p.lo, p.hi, p.b0, p.nb, it.cnt, it.wgoff, blockIdx.x - it.tile0);
}

template <int K>
__device__ __forceinline__ void dc_partition_body(const uint8_t *__restrict__ clean, uint64_t len, uint32_t lo, uint32_t hi,
                                                  uint32_t b0, uint32_t nb, const uint32_t *__restrict__ cnt,
                                                  const uint32_t *__restrict__ wgoff, uint32_t *__restrict__ base_out,
                                                  uint16_t *__restrict__ part, const uint32_t tile)
{
    extern __shared__ uint32_t dyn_lds[];                 // DP_LDS_BYTES
    uint32_t *stage = dyn_lds;                            // 64 KB: (bucket << 15 | value), bucketed
    uint32_t *h = stage + DT_TILE;                        // counts, then (start of this tile's range in the bucket) - (local start)
    uint16_t *lstart = reinterpret_cast<uint16_t *>(h + DC_MAX_NB);
    uint32_t *scan_lds = h + DC_MAX_NB + DC_MAX_NB / 2;
    for (uint32_t d = threadIdx.x; d < nb; d += DT_THREADS) h[d] = 0;
    __syncthreads();
    const uint64_t s = ((uint64_t)tile * DT_THREADS + threadIdx.x) * DT_SEG;
    Streams st;
    load_streams(st, clean, len, s);
    uint32_t wv[DT_SEG];   // word relative to the first bucket, or ~0
    uint16_t rk[DT_SEG];   // rank inside (tile, bucket)
    ForEachWindow<K, 0>::run(st, lo, hi, [&](int j, bool ok, uint32_t w) {
        wv[j] = 0xffffffffu;
        rk[j] = 0;
        if (ok) {
            wv[j] = w - (b0 << DC_VB);
            rk[j] = (uint16_t)atomicAdd(&h[wv[j] >> DC_VB], 1u);
        }
    });
    __syncthreads();
    // thread t owns buckets 4t .. 4t + 3: local starts, global bucket bases, this tile's offset in each bucket
    uint32_t ltot;
    {
        uint32_t c4[4], g4[4], lsum = 0, gsum = 0;
        const uint32_t d0 = threadIdx.x * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t d = d0 + e;
            c4[e] = d < nb ? h[d] : 0u;
            g4[e] = d < nb ? cnt[d] : 0u;
            lsum += c4[e];
            gsum += g4[e];
        }
        uint32_t gtot;
        uint32_t lex = block_scan512(lsum, &ltot, scan_lds);
        uint32_t gex = block_scan512(gsum, &gtot, scan_lds);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t d = d0 + e;
            if (d < nb) {
                lstart[d] = (uint16_t)lex;
                h[d] = gex + (c4[e] ? wgoff[(uint64_t)tile * nb + d] : 0u) - lex;
                if (tile == 0) base_out[d] = gex;
            }
            lex += c4[e];
            gex += g4[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DT_SEG; j++)
        if (wv[j] != 0xffffffffu) stage[(uint32_t)lstart[wv[j] >> DC_VB] + rk[j]] = wv[j];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ltot; i += DT_THREADS) {
        const uint32_t e = stage[i];
        part[(size_t)(uint32_t)(h[e >> DC_VB] + i)] = (uint16_t)(e & (DC_VALS - 1));
    }
}

template <int K>
__global__ __launch_bounds__(DT_THREADS) void dc_partition_kernel(const uint8_t *__restrict__ clean, uint64_t len, uint32_t lo,
                                                                   uint32_t hi, uint32_t b0, uint32_t nb,
                                                                   const uint32_t *__restrict__ cnt,
                                                                   const uint32_t *__restrict__ wgoff,
                                                                   uint32_t *__restrict__ base_out, uint16_t *__restrict__ part)
{
    dc_partition_body<K>(clean, len, lo, hi, b0, nb, cnt, wgoff, base_out, part, blockIdx.x);
}
template <int K>
__global__ __launch_bounds__(DT_THREADS) void dc_partition_batch_kernel(const DcBatch p)
{
    const uint32_t s = dc_batch_sample(p, blockIdx.x);
    const DcItem &it = p.it[s];
    dc_partition_body<K>(it.clean, it.len, p.lo, p.hi, p.b0, p.nb, it.cnt, it.wgoff, it.base, it.part, blockIdx.x - it.tile0);
}

// One workgroup per bucket.  bitmap: this sample's, DC_BUCKET_WORDS u64 per bucket.  mt_w / mt_f: multi-count
// entries before compaction; bucket b writes from slot base[b] / 2 on (it has at most cnt[b] / 2 of them, and the
// buckets' key ranges are laid out back to back, so the slots of different buckets cannot meet).
__global__ __launch_bounds__(DT_THREADS) void dc_count_kernel(const uint16_t *__restrict__ part, const uint32_t *__restrict__ cnt,
                                                               const uint32_t *__restrict__ base, uint32_t b0,
                                                               uint64_t *__restrict__ bitmap, uint32_t *__restrict__ mt_w,
                                                               uint32_t *__restrict__ mt_f, uint32_t *__restrict__ uniq_out,
                                                               uint32_t *__restrict__ multi_out, const uint32_t *__restrict__ need)
{
    extern __shared__ uint32_t dyn_lds[];   // DCNT_LDS_BYTES
    uint32_t *tbl = dyn_lds;                // 64 KB
    uint32_t *scan_lds = tbl + DC_VALS / 2;
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (need && !need[b]) return;   // second pass behind dc_count_sparse_kernel: only the buckets it left
    const uint32_t n = cnt[b];
    const size_t off = base[b];
    const uint32_t word0 = (b0 + b) << DC_VB;
    uint32_t uniq = 0, multi_total = 0;
    if (n < 65536u) {
        // two 16-bit counters per LDS word (no counter can overflow: the bucket has fewer than 65,536 keys); value v
        // lives in word ((v >> 1) & 31) * 512 + (v >> 6), so that the 32 words of thread t = v >> 6 are read without
        // bank conflicts
#pragma unroll
        for (int j = 0; j < 32; j++) tbl[j * DT_THREADS + t] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 8 * DT_THREADS) {   // eight loads in flight per thread
            uint32_t v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t i = i0 + e * DT_THREADS + t;
                v[e] = i < n ? (uint32_t)part[off + i] : 0xffffffffu;
            }
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (v[e] != 0xffffffffu) atomicAdd(&tbl[((v[e] >> 1) & 31u) * DT_THREADS + (v[e] >> 6)], 1u << ((v[e] & 1u) * 16));
        }
        __syncthreads();
        uint32_t w[32], multi = 0;
        uint64_t bits = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            w[j] = tbl[j * DT_THREADS + t];
            const uint32_t c0 = w[j] & 0xffffu, c1 = w[j] >> 16;
            if (c0) bits |= 1ull << (2 * j);
            if (c1) bits |= 1ull << (2 * j + 1);
            multi += (c0 >= 2) + (c1 >= 2);
        }
        uniq = (uint32_t)__popcll(bits);
        bitmap[(size_t)b * DC_BUCKET_WORDS + t] = bits;
        uint32_t dst = (uint32_t)(off / 2) + block_scan512(multi, &multi_total, scan_lds);
        if (multi) {
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const uint32_t c0 = w[j] & 0xffffu, c1 = w[j] >> 16;
                if (c0 >= 2) { mt_w[dst] = word0 + 64 * t + 2 * j; mt_f[dst] = c0; dst++; }
                if (c1 >= 2) { mt_w[dst] = word0 + 64 * t + 2 * j + 1; mt_f[dst] = c1; dst++; }
            }
        }
    } else {
        // 32-bit counters, the bucket's values in two halves of 2^14; value vv of a half lives in word
        // (vv & 63) * 256 + (vv >> 6)
        uint32_t mbase = (uint32_t)(off / 2);
        for (uint32_t half = 0; half < 2; half++) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 32; j++) tbl[j * DT_THREADS + t] = 0;
            __syncthreads();
            for (uint32_t i = t; i < n; i += DT_THREADS) {
                const uint32_t v = part[off + i];
                if ((v >> 14) == half) {
                    const uint32_t vv = v & 16383u;
                    atomicAdd(&tbl[(vv & 63u) * 256u + (vv >> 6)], 1u);
                }
            }
            __syncthreads();
            uint32_t multi = 0;
            uint64_t bits = 0;
            if (t < 256) {
                for (int j = 0; j < 64; j++) {
                    const uint32_t c = tbl[j * 256 + t];
                    if (c) bits |= 1ull << j;
                    multi += (c >= 2);
                }
                uniq += (uint32_t)__popcll(bits);
                bitmap[(size_t)b * DC_BUCKET_WORDS + half * 256 + t] = bits;
            }
            uint32_t mt;
            uint32_t dst = mbase + block_scan512(multi, &mt, scan_lds);
            if (multi) {
                for (int j = 0; j < 64; j++) {
                    const uint32_t c = tbl[j * 256 + t];
                    if (c >= 2) { mt_w[dst] = word0 + half * 16384u + 64 * t + j; mt_f[dst] = c; dst++; }
                }
            }
            mbase += mt;
            multi_total += mt;
        }
    }
    uint32_t utot;
    block_scan512(uniq, &utot, scan_lds);
    if (t == 0) { uniq_out[b] = utot; multi_out[b] = multi_total; }
}

// The same outputs for a bucket with few keys (a genome: 2,400 keys in 32,768 values, 7 % of them repeats) at a
// sixteenth of the LDS: instead of a counter per value, a presence bit per value (atomicOr; a key that finds its bit
// set is a repeat), a second bit set for the values seen twice, and counters only for those -- indexed by the rank of
// the value among the twice-seen ones, which the repeats increment.  8 workgroups per CU instead of 2, and 4 KB
// instead of 64 KB to clear and read back per bucket.  A bucket with more than SP_MAXDUP repeats (or SP_MAXKEYS keys)
// is left to dc_count_kernel: need[b] = 1.
constexpr int SP_THREADS = 256;
constexpr uint32_t SP_MAXDUP = 2048, SP_MAXKEYS = 16384;

__device__ __forceinline__ void dc_count_sparse_body(const uint16_t *__restrict__ part, const uint32_t *__restrict__ cnt,
                                                     const uint32_t *__restrict__ base, uint32_t b0, uint64_t *__restrict__ bitmap,
                                                     uint32_t *__restrict__ mt_w, uint32_t *__restrict__ mt_f,
                                                     uint32_t *__restrict__ uniq_out, uint32_t *__restrict__ multi_out,
                                                     uint32_t *__restrict__ need, const uint32_t bucket)
{
    __shared__ uint32_t seen[DC_VALS / 32], dup[DC_VALS / 32], rankb[DC_VALS / 32];   // 3 x 4 KB
    __shared__ uint32_t dcnt[SP_MAXDUP];     // occurrences beyond the first, by rank of the value among the repeated ones
    __shared__ uint16_t dlist[SP_MAXDUP];    // the repeats themselves
    __shared__ uint32_t scan_lds[SP_THREADS / 64];
    __shared__ uint32_t n_dup;
    const uint32_t b = bucket, t = threadIdx.x;
    const uint32_t n = cnt[b];
    const size_t off = base[b];
    if (n > SP_MAXKEYS) {
        if (t == 0) { need[b] = 1; uniq_out[b] = 0; multi_out[b] = 0; }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { seen[e * SP_THREADS + t] = 0; dup[e * SP_THREADS + t] = 0; }
#pragma unroll
    for (int e = 0; e < (int)(SP_MAXDUP / SP_THREADS); e++) dcnt[e * SP_THREADS + t] = 0;
    if (t == 0) n_dup = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 8 * SP_THREADS) {   // eight loads in flight per thread
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t i = i0 + e * SP_THREADS + t;
            v[e] = i < n ? (uint32_t)part[off + i] : 0xffffffffu;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (v[e] == 0xffffffffu) continue;
            const uint32_t bit = 1u << (v[e] & 31u);
            if (atomicOr(&seen[v[e] >> 5], bit) & bit) {
                atomicOr(&dup[v[e] >> 5], bit);
                const uint32_t pos = atomicAdd(&n_dup, 1u);
                if (pos < SP_MAXDUP) dlist[pos] = (uint16_t)v[e];
            }
        }
    }
    __syncthreads();
    const uint32_t nd = n_dup;
    if (nd > SP_MAXDUP) {   // uniform
        if (t == 0) { need[b] = 1; uniq_out[b] = 0; multi_out[b] = 0; }
        return;
    }
    // thread t owns words 4t .. 4t + 3 of the bit tables = values 128 t .. 128 t + 127 = bitmap words 2t, 2t + 1
    uint32_t sw[4], dw[4], dsum = 0, usum = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        sw[e] = seen[4 * t + e];
        dw[e] = dup[4 * t + e];
        usum += __popc(sw[e]);
        dsum += __popc(dw[e]);
    }
    uint32_t multi_total, utot;
    uint32_t rk = psk_block_excl_scan_u32<SP_THREADS>(dsum, &multi_total, scan_lds);
    psk_block_excl_scan_u32<SP_THREADS>(usum, &utot, scan_lds);
#pragma unroll
    for (int e = 0; e < 4; e++) { rankb[4 * t + e] = rk; rk += __popc(dw[e]); }
    reinterpret_cast<ulonglong2 *>(bitmap + (size_t)b * DC_BUCKET_WORDS)[t] =
        make_ulonglong2((uint64_t)sw[0] | ((uint64_t)sw[1] << 32), (uint64_t)sw[2] | ((uint64_t)sw[3] << 32));
    __syncthreads();
    for (uint32_t i = t; i < nd; i += SP_THREADS) {
        const uint32_t v = dlist[i];
        atomicAdd(&dcnt[rankb[v >> 5] + __popc(dup[v >> 5] & ((1u << (v & 31u)) - 1u))], 1u);
    }
    __syncthreads();
    if (dsum) {
        const uint32_t word0 = (b0 + b) << DC_VB;
        uint32_t r = rankb[4 * t];
        const uint32_t dst = (uint32_t)(off / 2);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint32_t m = dw[e];
            while (m) {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                mt_w[dst + r] = word0 + 128 * t + 32 * e + j;
                mt_f[dst + r] = 1u + dcnt[r];
                r++;
            }
        }
    }
    if (t == 0) { uniq_out[b] = utot; multi_out[b] = multi_total; need[b] = 0; }
}
__global__ __launch_bounds__(SP_THREADS) void dc_count_sparse_kernel(const uint16_t *__restrict__ part, const uint32_t *__restrict__ cnt,
                                                                      const uint32_t *__restrict__ base, uint32_t b0,
                                                                      uint64_t *__restrict__ bitmap, uint32_t *__restrict__ mt_w,
                                                                      uint32_t *__restrict__ mt_f, uint32_t *__restrict__ uniq_out,
                                                                      uint32_t *__restrict__ multi_out, uint32_t *__restrict__ need)
{
    dc_count_sparse_body(part, cnt, base, b0, bitmap, mt_w, mt_f, uniq_out, multi_out, need, blockIdx.x);
}
__global__ __launch_bounds__(SP_THREADS) void dc_count_sparse_batch_kernel(const DcBatch p)
{
    const DcItem &it = p.it[blockIdx.y];
    dc_count_sparse_body(it.part, it.cnt, it.base, p.b0, it.bitmap, it.mt_w, it.mt_f, it.uniq, it.multi, it.need, blockIdx.x);
}

// totals[0] = keys kept, [1] = distinct words, [2] = multi-count entries, [3] = buckets the sparse pass left to the
// table pass -- also written straight into pinned host
// memory (`host_totals`), which saves the copy kernel of a 16-byte device-to-host transfer
__device__ __forceinline__ void dc_totals_body(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ uniq,
                                               const uint32_t *__restrict__ multi, const uint32_t *__restrict__ need, uint32_t nb,
                                               uint32_t *__restrict__ totals, uint32_t *__restrict__ host_totals)
{
    __shared__ uint32_t lds[4][16];
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    uint32_t c = 0, u = 0, m = 0, nd = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t d = e * 1024 + t;
        if (d < nb) { c += cnt[d]; u += uniq[d]; m += multi[d]; nd += need ? need[d] : 0u; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c += __shfl_xor(c, d, 64);
        u += __shfl_xor(u, d, 64);
        m += __shfl_xor(m, d, 64);
        nd += __shfl_xor(nd, d, 64);
    }
    if (lane == 0) { lds[0][wid] = c; lds[1][wid] = u; lds[2][wid] = m; lds[3][wid] = nd; }
    __syncthreads();
    if (t < 4) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) s += lds[t][w];
        totals[t] = s;
        host_totals[t] = s;
    }
}

__global__ __launch_bounds__(1024) void dc_totals_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ uniq,
                                                         const uint32_t *__restrict__ multi, const uint32_t *__restrict__ need,
                                                         uint32_t nb, uint32_t *__restrict__ totals,
                                                         uint32_t *__restrict__ host_totals)
{
    dc_totals_body(cnt, uniq, multi, need, nb, totals, host_totals);
}
__global__ __launch_bounds__(1024) void dc_totals_batch_kernel(const DcBatch p)
{
    const DcItem &it = p.it[blockIdx.x];
    dc_totals_body(it.cnt, it.uniq, it.multi, it.need, p.nb, it.totals, it.host_totals);
}

// bucket b's multi-count entries -> their place in the arena block (offset = sum of multi[0 .. b))
__device__ __forceinline__ void dc_compact_body(const uint32_t *__restrict__ mt_w, const uint32_t *__restrict__ mt_f,
                                                const uint32_t *__restrict__ base, const uint32_t *__restrict__ multi,
                                                uint32_t *__restrict__ dst_w, uint32_t *__restrict__ dst_f, const uint32_t bucket)
{
    __shared__ uint32_t lds[4];
    const uint32_t b = bucket, t = threadIdx.x;
    uint32_t below = 0;
#pragma unroll
    for (int e = 0; e < (int)(DC_MAX_NB / 256); e++) {
        const uint32_t d = e * 256 + t;
        if (d < b) below += multi[d];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d, 64);
    if ((t & 63) == 0) lds[t >> 6] = below;
    __syncthreads();
    const uint32_t dst = lds[0] + lds[1] + lds[2] + lds[3];
    const uint32_t n = multi[b], src = base[b] / 2;
    for (uint32_t i = t; i < n; i += 256) {
        dst_w[dst + i] = mt_w[src + i];
        dst_f[dst + i] = mt_f[src + i];
    }
}
__global__ __launch_bounds__(256) void dc_compact_kernel(const uint32_t *__restrict__ mt_w, const uint32_t *__restrict__ mt_f,
                                                         const uint32_t *__restrict__ base, const uint32_t *__restrict__ multi,
                                                         uint32_t *__restrict__ dst_w, uint32_t *__restrict__ dst_f)
{
    dc_compact_body(mt_w, mt_f, base, multi, dst_w, dst_f, blockIdx.x);
}
__global__ __launch_bounds__(256) void dc_compact_batch_kernel(const DcBatch p)
{
    const DcItem &it = p.it[blockIdx.y];
    dc_compact_body(it.mt_w, it.mt_f, it.base, it.multi, it.dst_w, it.dst_f, blockIdx.x);
}

// ---- dense form -> words[] / freqs[] ---------------------------------------------------------------------------
__global__ void dm_popc_kernel(const uint64_t *__restrict__ bitmap, uint64_t n_words, uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) out[i] = (uint32_t)__popcll(bitmap[i]);
}

__global__ void dm_expand_kernel(const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ rank, uint64_t n_words,
                                 uint64_t word0, uint64_t *__restrict__ words, uint32_t *__restrict__ freqs)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    uint64_t m = bitmap[i];
    uint32_t o = rank[i];
    while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        words[o] = word0 + i * 64 + j;
        freqs[o] = 1u;
        o++;
    }
}

__global__ void dm_multi_kernel(const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ rank, uint64_t word0,
                                const uint32_t *__restrict__ mwords, const uint32_t *__restrict__ mfreqs, uint64_t n_multi,
                                uint32_t *__restrict__ freqs)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_multi) return;
    const uint64_t v = (uint64_t)mwords[i] - word0;
    const uint64_t below = bitmap[v >> 6] & ((1ull << (v & 63)) - 1ull);
    freqs[rank[v >> 6] + (uint32_t)__popcll(below)] = mfreqs[i];
}

__global__ void dense_lookup_kernel(const uint64_t *__restrict__ bitmap, uint64_t word0, uint64_t n_vals,
                                    const uint32_t *__restrict__ mwords, const uint32_t *__restrict__ mfreqs, uint64_t n_multi,
                                    const uint64_t *__restrict__ q, uint64_t n, uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = q[i];
    uint32_t f = 0;
    if (key >= word0 && key - word0 < n_vals) {
        const uint64_t v = key - word0;
        if ((bitmap[v >> 6] >> (v & 63)) & 1ull) {
            f = 1;
            uint64_t lo = 0, hi = n_multi;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (mwords[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < n_multi && mwords[lo] == key) f = mfreqs[lo];
        }
    }
    out[i] = f;
}

}  // namespace

void dense_configure(psk_ctx *ctx)
{
    ctx->dense_mode = false;
    ctx->dense_b0 = ctx->dense_nb = 0;
    if (getenv("PSK_NO_DENSE")) return;
    const int k = ctx->k;
    // below 2^21 words a bucket holds too large a share of a sample's keys for one workgroup; above 2^26 the bitmap
    // outgrows the list it replaces
    if (2 * k > 26 || 2 * k < 21) return;
    const uint64_t space = 1ull << (2 * k);
    const uint64_t lo = ctx->slab_lo, hi = (ctx->slab_hi && ctx->slab_hi < space) ? ctx->slab_hi : space;
    if (lo >= hi) return;
    const uint32_t b0 = (uint32_t)(lo >> DC_VB), b1 = (uint32_t)((hi - 1) >> DC_VB);
    const uint32_t nb = b1 - b0 + 1;
    if (nb > DC_MAX_NB) return;
    if ((size_t)ctx->n_samples * nb * DC_BUCKET_WORDS * 8 > DC_DENSE_BYTES_MAX) return;
    ctx->dense_mode = true;
    ctx->dense_b0 = b0;
    ctx->dense_nb = nb;
}

namespace {

struct DcBufs {
    uint32_t *cnt, *base, *uniq, *multi, *need, *totals, *mt_w, *mt_f;
};

DcBufs dc_bufs(const CountLane &L, uint32_t slot)
{
    uint32_t *meta = L.dc_meta.as<uint32_t>();
    DcBufs d;
    d.cnt = L.dc_cnt.as<uint32_t>() + (size_t)DC_MAX_NB * slot;
    d.base = meta;
    d.uniq = meta + DC_MAX_NB;
    d.multi = meta + 2 * DC_MAX_NB;
    d.need = meta + 3 * DC_MAX_NB;
    d.totals = meta + 4 * DC_MAX_NB;
    d.mt_w = L.dc_mtemp.as<uint32_t>();
    d.mt_f = d.mt_w + (L.n / 2 + 2);
    return d;
}

template <int K>
int launch_tiles(psk_ctx *ctx, const CountLane &L, const DcBufs &d, uint64_t clean_len, uint32_t n_tiles, uint32_t lo, uint32_t hi)
{
    static PerDeviceOnce lds_set;
    if (lds_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(dc_partition_kernel<K>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)DP_LDS_BYTES));
    }
    const uint8_t *clean = L.raw.as<uint8_t>();
    dc_hist_kernel<K><<<n_tiles, DT_THREADS, 0, ctx->stream>>>(clean, clean_len, lo, hi, ctx->dense_b0, ctx->dense_nb, d.cnt,
                                                               L.dc_wgoff.as<uint32_t>());
    PSK_HIP(ctx, hipGetLastError());
    dc_partition_kernel<K><<<n_tiles, DT_THREADS, DP_LDS_BYTES, ctx->stream>>>(clean, clean_len, lo, hi, ctx->dense_b0, ctx->dense_nb,
                                                                              d.cnt, L.dc_wgoff.as<uint32_t>(), d.base,
                                                                              L.dc_part.as<uint16_t>());
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

int launch_table_count(psk_ctx *ctx, const CountLane &L, const DcBufs &d, uint64_t *bitmap, const uint32_t *need)
{
    static PerDeviceOnce lds_set;
    if (lds_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(dc_count_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)DCNT_LDS_BYTES));
    }
    dc_count_kernel<<<ctx->dense_nb, DT_THREADS, DCNT_LDS_BYTES, ctx->stream>>>(L.dc_part.as<uint16_t>(), d.cnt, d.base, ctx->dense_b0,
                                                                               bitmap, d.mt_w, d.mt_f, d.uniq, d.multi, need);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

}  // namespace

int dense_chain_enqueue(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n)
{
    SampleList &S = ctx->lists[sample_idx];
    const uint32_t nb = ctx->dense_nb;
    const uint64_t space = 1ull << (2 * ctx->k);
    const uint32_t lo = (uint32_t)ctx->slab_lo;
    const uint32_t hi = (uint32_t)((ctx->slab_hi && ctx->slab_hi < space) ? ctx->slab_hi : space);
    S.dense = true;
    PSK_TRY(arena_alloc(ctx, (size_t)nb * DC_BUCKET_WORDS * 8, (void **)&S.bitmap));
    const uint32_t n_tiles = div_up(clean_len, DT_TILE);
    PSK_TRY(dev_reserve(ctx, L.dc_part, n * 2 + 64));
    PSK_TRY(dev_reserve(ctx, L.dc_wgoff, (size_t)n_tiles * nb * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_cnt, (size_t)DC_SLOTS * DC_MAX_NB * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_meta, (size_t)(4 * DC_MAX_NB + 4) * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_mtemp, (n / 2 + 2) * 8));
    if (L.dc_slot == 0 || L.dc_slot >= DC_SLOTS) {
        PSK_HIP(ctx, hipMemsetAsync(L.dc_cnt.p, 0, (size_t)DC_SLOTS * DC_MAX_NB * 4, ctx->stream));
        L.dc_slot = 0;
    }
    const DcBufs d = dc_bufs(L, L.dc_slot++);
    switch (ctx->k) {
    case 11: PSK_TRY(launch_tiles<11>(ctx, L, d, clean_len, n_tiles, lo, hi)); break;
    case 12: PSK_TRY(launch_tiles<12>(ctx, L, d, clean_len, n_tiles, lo, hi)); break;
    case 13: PSK_TRY(launch_tiles<13>(ctx, L, d, clean_len, n_tiles, lo, hi)); break;
    default: return psk_fail(ctx, PSK_ESTATE, "dense counting is built for k = 11..13, not %d", ctx->k);
    }
    PSK_HIP(ctx, hipEventRecord(L.raw_free, ctx->stream));
    L.raw_used = true;
    // few keys per bucket (a genome): the presence-bit pass, which leaves over-full buckets to the table pass that
    // dense_chain_finalize runs when the totals report any; many (reads at depth): the table pass at once
    const bool sparse = n / nb < 4096 && !getenv("PSK_DC_TABLE");
    if (sparse)
        dc_count_sparse_kernel<<<nb, SP_THREADS, 0, ctx->stream>>>(L.dc_part.as<uint16_t>(), d.cnt, d.base, ctx->dense_b0, S.bitmap,
                                                                   d.mt_w, d.mt_f, d.uniq, d.multi, d.need);
    else
        PSK_TRY(launch_table_count(ctx, L, d, S.bitmap, nullptr));
    PSK_HIP(ctx, hipGetLastError());
    dc_totals_kernel<<<1, 1024, 0, ctx->stream>>>(d.cnt, d.uniq, d.multi, sparse ? d.need : nullptr, nb, d.totals, L.pinned_cnt);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipEventRecord(L.done, ctx->stream));
    L.dense = true;
    return PSK_OK;
}

// after hipEventSynchronize(L.done): sizes and fills the arena block of the multi-count entries
int dense_chain_finalize(psk_ctx *ctx, CountLane &L, uint64_t *n_kept, uint64_t *n_unique)
{
    SampleList &S = ctx->lists[L.sample];
    const DcBufs d = dc_bufs(L, L.dc_slot - 1);
    if (L.pinned_cnt[3]) {
        // some buckets were too full for the presence-bit pass: the table pass on those, then the totals again (the
        // sample's buffers are untouched until the sample after next is queued on this set)
        PSK_TRY(launch_table_count(ctx, L, d, S.bitmap, d.need));
        dc_totals_kernel<<<1, 1024, 0, ctx->stream>>>(d.cnt, d.uniq, d.multi, nullptr, ctx->dense_nb, d.totals, L.pinned_cnt);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_kept = L.pinned_cnt[0];
    *n_unique = L.pinned_cnt[1];
    const uint64_t nm = L.pinned_cnt[2];
    S.n_multi = nm;
    if (nm) {
        PSK_TRY(arena_alloc(ctx, nm * 4, (void **)&S.mwords));
        PSK_TRY(arena_alloc(ctx, nm * 4, (void **)&S.mfreqs));
        if (!L.dc_defer_compact) {
            dc_compact_kernel<<<ctx->dense_nb, 256, 0, ctx->stream>>>(d.mt_w, d.mt_f, d.base, d.multi, S.mwords, S.mfreqs);
            PSK_HIP(ctx, hipGetLastError());
        }
    }
    L.dense = false;
    return PSK_OK;
}

// ---- a group of samples in one launch chain (psk_count_kmers_batch, genomes at k = 11..13) -------------------------------
int dense_group_size()
{
    const char *e = getenv("PSK_DC_GROUP");   // read per call: tests cross group sizes in one process
    const int v = e ? atoi(e) : DC_GROUP;
    return v < 1 ? 1 : (v > DC_GROUP ? DC_GROUP : v);
}

void dense_lane_bytes(const psk_ctx *ctx, size_t max_len, size_t out[5])
{
    out[0] = max_len * 2 + 64;                                           // dc_part
    out[1] = (size_t)div_up(max_len, DT_TILE) * ctx->dense_nb * 4;       // dc_wgoff
    out[2] = (size_t)DC_SLOTS * DC_MAX_NB * 4;                           // dc_cnt
    out[3] = (size_t)(4 * DC_MAX_NB + 4) * 4;                            // dc_meta
    out[4] = (max_len / 2 + 2) * 8;                                      // dc_mtemp
}

// may this sample's chain ride in a group?  (few keys per bucket: the presence-bit pass; read sets take the table pass alone)
bool dense_group_ok(const psk_ctx *ctx, uint64_t n) { return ctx->dense_mode && n > 0 && n / ctx->dense_nb < 4096 && !getenv("PSK_DC_TABLE"); }

template <int K>
static int launch_group_tiles(psk_ctx *ctx, const DcBatch &p, uint32_t tiles)
{
    static PerDeviceOnce lds_set;
    if (lds_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(dc_partition_batch_kernel<K>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)DP_LDS_BYTES));
    }
    dc_hist_batch_kernel<K><<<tiles, DT_THREADS, 0, ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    dc_partition_batch_kernel<K><<<tiles, DT_THREADS, DP_LDS_BYTES, ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

// The chains of `count` samples (each on its own buffer set, clean stream resident, dense_group_ok) as ONE chain: what
// dense_chain_enqueue does per sample, with every launch covering the group.  Each set's events fire as for its own chain.
int dense_group_enqueue(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, const uint64_t *clean_len, const uint64_t *n,
                        int count)
{
    if (count < 1 || count > DC_GROUP) return psk_fail(ctx, PSK_EINVAL, "bad group size %d", count);
    const uint32_t nb = ctx->dense_nb;
    const uint64_t space = 1ull << (2 * ctx->k);
    DcBatch p;
    memset(&p, 0, sizeof(p));
    p.n = (uint32_t)count;
    p.lo = (uint32_t)ctx->slab_lo;
    p.hi = (uint32_t)((ctx->slab_hi && ctx->slab_hi < space) ? ctx->slab_hi : space);
    p.b0 = ctx->dense_b0;
    p.nb = nb;
    uint32_t tiles = 0;
    for (int s = 0; s < count; s++) {
        CountLane &L = *lanes[s];
        SampleList &S = ctx->lists[sample_idx[s]];
        S.dense = true;
        PSK_TRY(arena_alloc(ctx, (size_t)nb * DC_BUCKET_WORDS * 8, (void **)&S.bitmap));
        const uint32_t n_tiles = div_up(clean_len[s], DT_TILE);
        PSK_TRY(dev_reserve(ctx, L.dc_part, n[s] * 2 + 64));
        PSK_TRY(dev_reserve(ctx, L.dc_wgoff, (size_t)n_tiles * nb * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_cnt, (size_t)DC_SLOTS * DC_MAX_NB * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_meta, (size_t)(4 * DC_MAX_NB + 4) * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_mtemp, (n[s] / 2 + 2) * 8));
        if (L.dc_slot == 0 || L.dc_slot >= DC_SLOTS) {
            PSK_HIP(ctx, hipMemsetAsync(L.dc_cnt.p, 0, (size_t)DC_SLOTS * DC_MAX_NB * 4, ctx->stream));
            L.dc_slot = 0;
        }
        const DcBufs d = dc_bufs(L, L.dc_slot++);
        DcItem &it = p.it[s];
        it.clean = L.raw.as<uint8_t>();
        it.len = clean_len[s];
        it.cnt = d.cnt; it.wgoff = L.dc_wgoff.as<uint32_t>(); it.base = d.base;
        it.part = L.dc_part.as<uint16_t>();
        it.bitmap = S.bitmap;
        it.mt_w = d.mt_w; it.mt_f = d.mt_f; it.uniq = d.uniq; it.multi = d.multi; it.need = d.need; it.totals = d.totals;
        it.host_totals = L.pinned_cnt;
        it.tile0 = tiles;
        tiles += n_tiles;
    }
    switch (ctx->k) {
    case 11: PSK_TRY(launch_group_tiles<11>(ctx, p, tiles)); break;
    case 12: PSK_TRY(launch_group_tiles<12>(ctx, p, tiles)); break;
    case 13: PSK_TRY(launch_group_tiles<13>(ctx, p, tiles)); break;
    default: return psk_fail(ctx, PSK_ESTATE, "dense counting is built for k = 11..13, not %d", ctx->k);
    }
    for (int s = 0; s < count; s++) {
        PSK_HIP(ctx, hipEventRecord(lanes[s]->raw_free, ctx->stream));
        lanes[s]->raw_used = true;
    }
    dc_count_sparse_batch_kernel<<<dim3(nb, (uint32_t)count), SP_THREADS, 0, ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    dc_totals_batch_kernel<<<(uint32_t)count, 1024, 0, ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipEventRecord(lanes[count - 1]->done, ctx->stream));
    for (int s = 0; s < count; s++) {
        if (s + 1 < count) PSK_HIP(ctx, hipEventRecord(lanes[s]->done, ctx->stream));
        lanes[s]->dense = true;
        lanes[s]->dc_defer_compact = true;
    }
    return PSK_OK;
}

// after chain_finalize of every set of the group (which sized and allocated the multi-count blocks): ONE compaction launch
int dense_group_compact(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, int count)
{
    DcBatch p;
    memset(&p, 0, sizeof(p));
    p.nb = ctx->dense_nb;
    uint32_t m = 0;
    for (int s = 0; s < count; s++) {
        CountLane &L = *lanes[s];
        L.dc_defer_compact = false;
        const SampleList &S = ctx->lists[sample_idx[s]];
        if (!S.dense || !S.n_multi) continue;
        const DcBufs d = dc_bufs(L, L.dc_slot - 1);
        DcItem &it = p.it[m++];
        it.mt_w = d.mt_w; it.mt_f = d.mt_f; it.base = d.base; it.multi = d.multi;
        it.dst_w = S.mwords; it.dst_f = S.mfreqs;
    }
    p.n = m;
    if (m) {
        dc_compact_batch_kernel<<<dim3(ctx->dense_nb, m), 256, 0, ctx->stream>>>(p);
        PSK_HIP(ctx, hipGetLastError());
    }
    return PSK_OK;
}

int dense_materialize(psk_ctx *ctx, int first, int n)
{
    for (int i = first; i < first + n; i++) {
        SampleList &S = ctx->lists[i];
        if (!S.done || !S.dense || S.words || S.n_unique == 0) continue;
        const uint64_t nw = (uint64_t)ctx->dense_nb * DC_BUCKET_WORDS;
        const uint64_t word0 = (uint64_t)ctx->dense_b0 << DC_VB;
        PSK_TRY(dev_reserve(ctx, ctx->flags, nw * 4));
        uint32_t *rank = ctx->flags.as<uint32_t>();
        dm_popc_kernel<<<div_up(nw, 256), 256, 0, ctx->stream>>>(S.bitmap, nw, rank);
        PSK_HIP(ctx, hipGetLastError());
        PSK_TRY(dev_exclusive_scan_u32(ctx, rank, rank, nw, nullptr));
        PSK_TRY(arena_alloc(ctx, S.n_unique * 8, (void **)&S.words));
        PSK_TRY(arena_alloc(ctx, S.n_unique * 4, (void **)&S.freqs));
        dm_expand_kernel<<<div_up(nw, 256), 256, 0, ctx->stream>>>(S.bitmap, rank, nw, word0, S.words, S.freqs);
        PSK_HIP(ctx, hipGetLastError());
        if (S.n_multi) {
            dm_multi_kernel<<<div_up(S.n_multi, 256), 256, 0, ctx->stream>>>(S.bitmap, rank, word0, S.mwords, S.mfreqs, S.n_multi,
                                                                           S.freqs);
            PSK_HIP(ctx, hipGetLastError());
        }
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `rank` is reused by the next sample
    }
    return PSK_OK;
}

int dense_lookup_counts(psk_ctx *ctx, const SampleList &L, const uint64_t *d_query, uint64_t n, uint32_t *d_out)
{
    dense_lookup_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(L.bitmap, (uint64_t)ctx->dense_b0 << DC_VB,
                                                                 (uint64_t)ctx->dense_nb << DC_VB, L.mwords, L.mfreqs, L.n_multi,
                                                                 d_query, n, d_out);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}
