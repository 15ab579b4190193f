// a1 for the word spaces beyond the dense form (k = 14..32; r05: the 64-bit words of k = 17..32 as well -- every kernel below is a
// template over the word type W = uint32_t (k <= 16, 512-thread tiles of 16,384 bases) | uint64_t (k >= 17, 256-thread tiles of
// 8,192 bases: a tile's staged words are 64 KB of LDS either way)): a bucketed sort instead of four (eight) radix passes
// (replaces bin/glistmaker; modeling.py:303-315).  The radix route runs 17 launches per sample (extract, 4 x
// (histogram, digit scan, scatter), 3 for the run lengths) -- 250-300 us for a 5-Mbp genome, and hardly less for the
// 600,000 words a rank keeps of it under an 8-way slab filter, because every launch has its floor.  Here the word space
// is cut into buckets of equal COUNT, about 2,400 words each (64..2,048 buckets) -- the splitters are quantiles of the
// first list of the run, so that an AT-rich genome fills them as evenly as a uniform one -- and a sample is counted by
//   bs_hist_kernel       tile of 16,384 bases per workgroup: canonical words (kmer_windows.h), slab filter, bucket from a
//                        coarse cell table + a short search over the splitters in LDS, histogram; one returning atomic per non-empty
//                        (tile, bucket) reserves the tile's range in the bucket
//   bs_partition_kernel  the same tile again: rank inside (tile, bucket) from a returning LDS atomic, the tile's words
//                        bucketed in LDS and spilled in bucket runs
//   bs_sort_kernel       a bucket at a time per workgroup: its ~2,400 words dealt into 2,048 sub-bins of its range in LDS,
//                        stretches of <= 16 words sorted in registers, run lengths from a scan of the run heads, unique words + counts
//                        written over the bucket's own key range
//   bs_totals_kernel     unique counts -> offsets, totals to the host
//   bs_compact_kernel    (one sample later, once the host has sized the arena block) packs words (u64) + counts
// The output is the ordinary sparse list.  A bucket that outgrows the LDS sort (a sample unlike the one the splitters
// came from) sends the sample through the radix sort of its partitioned words instead (bucket_chain_finalize).
// Integer work, LDS / HBM bound: no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"
#include "kmer_windows.h"

namespace {

// threads of a counting tile: 512 x 32 window ends = 16,384 bases for 32-bit words, 256 x 32 = 8,192 for 64-bit ones
template <typename W> struct BsGeo { static constexpr int THREADS = sizeof(W) == 4 ? 512 : 256; static constexpr int TILE = THREADS * KW_SEG; };
constexpr uint32_t BS_SLOTS = 32;                   // pre-zeroed counter slots per buffer set
// threads of a sorting workgroup.  r06: TWO workgroups per CU for either width -- a bucket's sort is a chain of short steps between
// barriers, and a second bucket in flight is what fills the waits (r05: the 64-bit instance held the bucket twice in 148 KB of LDS:
// one workgroup per CU; the 32-bit one was 256 bytes over half the CU's 160 KB, so it ran alone as well).  64-bit words: 512 threads
// (90 VGPRs: 2 x 8 waves per CU = 4 per SIMD), 32-bit: 512 threads, three workgroups per CU.
// (the 32-bit instance, 49 KB of LDS a workgroup: 512 threads x 3 per CU against 1,024 x 2 -- sort 356 -> 292 us per 8 genomes at k = 16
// whole-space, 78 -> 49 under a 1/8 slab; 512 x 2: 310 / 53)
#ifndef PSK_BS_SORT32_THREADS
#define PSK_BS_SORT32_THREADS 512
#endif
#ifndef PSK_BS_SORT32_WGS
#define PSK_BS_SORT32_WGS 3
#endif
template <typename W> struct SortGeo {
    static constexpr int THREADS = sizeof(W) == 4 ? PSK_BS_SORT32_THREADS : 512;
    static constexpr int WGS_PER_CU = sizeof(W) == 4 ? PSK_BS_SORT32_WGS : 2;
    static constexpr int WAVES_PER_SIMD = THREADS / 64 * WGS_PER_CU / 4;   // what the register budget has to allow
};
#ifndef PSK_BS_CAPMAX
#define PSK_BS_CAPMAX 8128
#endif
constexpr uint32_t BS_CAP_MAX = PSK_BS_CAPMAX;      // words a bucket may hold for the LDS sort (8,128 x 8 B + 16.1 KB = 81.5 KB: two per CU)
constexpr int BS_CELLS = 4096;                      // cells of the coarse table over the run's word range
// stage | splitters | h | lstart | scan | cells | buckets of the staged words   (32-bit words: 135 KB; 64-bit: 124 KB)
template <typename W> constexpr size_t bp_lds_bytes()
{
    return (size_t)BsGeo<W>::TILE * sizeof(W) + (size_t)BS_NB * sizeof(W) + (size_t)(BS_NB + BS_NB / 2 + 16 + BS_CELLS) * 4 + (size_t)BsGeo<W>::TILE * 2;
}
// the bucket's words in sub-bin order | sub-bin starts | fills | scan   (32-bit words: 49 KB; 64-bit: 81.5 KB: two workgroups per CU)
template <typename W> constexpr size_t bsort_lds_bytes() { return (size_t)BS_CAP_MAX * sizeof(W) + (size_t)(2 * 2048 + 32) * 4; }

// The bucket of a word = the largest b with spl[b] <= w (spl[0] = the first word of the slab).  A binary search over the splitters is eleven
// dependent LDS reads per window; a coarse table over 4,096 equal cells of the run's word range (ct[c] = bucket of the
// cell's first word) leaves a search over the one to three buckets that meet the cell.
template <typename W>
struct BsMap {
    W lo;             // first word of the range the cells cover (the slab's)
    uint32_t shift;   // cell = (w - lo) >> shift
    uint32_t nb;      // buckets in use (a power of two, 64..2048): about 2,400 words each
};

template <typename W>
__device__ __forceinline__ uint32_t bucket_search(const W *spl, uint32_t nb, W w)
{
    uint32_t b = 0;
    for (uint32_t step = nb >> 1; step > 0; step >>= 1)
        if (spl[b + step] <= w) b += step;
    return b;
}

// r06: the buckets of G windows at once.  One look-up is a chain of dependent LDS reads -- cell, then splitters -- and the counting
// kernels run ONE wave per SIMD (their LDS leaves room for one workgroup per CU), so nothing hides a read's latency but
// the thread's own independent work: 32 look-ups one after the other, each with its own loop, cost 3-4 exposed LDS round trips
// per window (bs_partition_batch_kernel<21>: 432 us per 8 genomes, 0.032 of the HBM roofline; hardly fewer under a slab filter that
// drops 7 windows of 8 -- it was never the atomics).  Here the cells of G windows are read together (ct2[c] = first bucket of cell
// c | first bucket of cell c + 1 << 16: one read), then the two splitters that decide a cell meeting up to three buckets -- both
// addresses known from the cell, so the 2 G reads are independent --, then a compare: two round trips per G windows.  A cell that
// meets more than three buckets (a run unlike the sample the splitters came from) takes the search loop afterwards.
#ifndef PSK_BS_G
#define PSK_BS_G 8
#endif
constexpr int BS_G = PSK_BS_G;
template <typename W, int G>
__device__ __forceinline__ void buckets_of(const W *spl, const uint32_t *ct2, const BsMap<W> &mp, const W *w, uint32_t *b)
{
    uint32_t be[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
        const W cw = (w[j] - mp.lo) >> mp.shift;   // (a word that is none -- all ones -- or outside the slab lands in some cell: harmless)
        be[j] = ct2[cw < (W)BS_CELLS ? (uint32_t)cw : (uint32_t)BS_CELLS - 1u];
    }
    W p1[G], p2[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
        const uint32_t b0 = be[j] & 0xffffu, e = be[j] >> 16;
        p1[j] = spl[b0 + 1u < e ? b0 + 1u : e];
        p2[j] = spl[b0 + 2u < e ? b0 + 2u : e];
    }
    bool far = false;
#pragma unroll
    for (int j = 0; j < G; j++) {
        const uint32_t b0 = be[j] & 0xffffu, e = be[j] >> 16;
        b[j] = b0 + ((b0 + 1u <= e && p1[j] <= w[j]) ? 1u : 0u) + ((b0 + 2u <= e && p2[j] <= w[j]) ? 1u : 0u);
        far = far || e > b0 + 2u;
    }
    if (__ballot(far)) {
#pragma unroll
        for (int j = 0; j < G; j++) {
            uint32_t lo = be[j] & 0xffffu, hi = be[j] >> 16;
            if (hi > lo + 2u) {
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (spl[mid] <= w[j]) lo = mid;
                    else hi = mid - 1u;
                }
                b[j] = lo;
            }
        }
    }
}

// splitters = every (nu / nb)-th word of a sorted list of this run; then the coarse table
template <typename W>
__global__ void bs_splitters_kernel(const uint64_t *__restrict__ words, uint64_t nu, uint32_t nb, W lo, W *__restrict__ spl)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= BS_NB) return;
    spl[b] = b == 0 ? lo :   // the first bucket begins where the slab does: a bucket's range is what its sub-bins divide
             (b < nb ? (W)words[(uint64_t)b * nu / nb] : (W)~(W)0);
}

template <typename W>
__global__ void bs_cells_kernel(const W *__restrict__ spl, BsMap<W> mp, uint16_t *__restrict__ ct)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > (uint32_t)BS_CELLS) return;
    // the first word of cell c, or "beyond the last word" (c << shift stays below 2^64: c < 2^12, shift <= 52)
    const uint64_t off = (uint64_t)c << mp.shift, first = (uint64_t)mp.lo + off;
    const bool beyond = c == (uint32_t)BS_CELLS || first < off || first > (uint64_t)(W)~(W)0;
    ct[c] = (uint16_t)(beyond ? mp.nb - 1u : bucket_search<W>(spl, mp.nb, (W)first));
}

// the workgroup's copies of the splitters and the cells (the cells as pairs: see buckets_of)
template <typename W>
__device__ __forceinline__ void load_map2(W *spl, uint32_t *ct2, const W *__restrict__ spl_g, const uint16_t *__restrict__ ct_g, int threads)
{
    for (uint32_t d = threadIdx.x; d < BS_NB; d += threads) spl[d] = spl_g[d];
    for (uint32_t d = threadIdx.x; d < (uint32_t)BS_CELLS; d += threads) ct2[d] = (uint32_t)ct_g[d] | ((uint32_t)ct_g[d + 1] << 16);
}

// ---- a group of samples per launch (bucket_group_enqueue; see dense_count.hip: a genome is one tile per CU) ---------------
constexpr int BS_GROUP = 8;
template <typename W>
struct BsItem {
    const uint8_t *clean;
    uint64_t len;
    uint32_t *cnt, *wgoff, *base, *ctmp, *uniq, *uoff, *flag, *host;
    W *part, *wtmp;
    uint64_t *words;
    uint32_t *freqs;
    uint32_t tile0;   // first tile of the sample in the group's grid
};
template <typename W>
struct BsBatch {
    uint32_t n, cap;
    W lo, hi, last_word;
    BsMap<W> mp;
    const W *spl;
    const uint16_t *ct;
    BsItem<W> it[BS_GROUP];
};
template <typename W>
__device__ __forceinline__ uint32_t bs_batch_sample(const BsBatch<W> &p, uint32_t tile)
{
    uint32_t s = 0;
#pragma unroll
    for (int q = 1; q < BS_GROUP; q++) s += (q < (int)p.n && tile >= p.it[q].tile0) ? 1u : 0u;
    return s;
}

template <int K>
__device__ __forceinline__ void bs_hist_body(const uint8_t *__restrict__ clean, uint64_t len, typename Windows<K>::Word lo,
                                             typename Windows<K>::Word hi, const typename Windows<K>::Word *__restrict__ spl_g,
                                             const uint16_t *__restrict__ ct_g, const BsMap<typename Windows<K>::Word> mp,
                                             uint32_t *__restrict__ cnt, uint32_t *__restrict__ wgoff, const uint32_t tile)
{
    typedef typename Windows<K>::Word W;
    constexpr int BT_THREADS = BsGeo<W>::THREADS;
    __shared__ uint32_t h[BS_NB];
    __shared__ W spl[BS_NB];
    __shared__ uint32_t ct2[BS_CELLS];
    constexpr W NONE = (W)~(W)0;                          // (a valid canonical word is never all ones: its reverse complement would be 0)
    for (uint32_t d = threadIdx.x; d < BS_NB; d += BT_THREADS) h[d] = 0;
    load_map2<W>(spl, ct2, spl_g, ct_g, BT_THREADS);
    __syncthreads();
    const uint64_t s = ((uint64_t)tile * BT_THREADS + threadIdx.x) * KW_SEG;
    typename Windows<K>::St st;
    load_streams(st, clean, len, s);
    W wv[KW_SEG];
    Windows<K>::run(st, lo, hi, [&](int j, bool ok, W w) { wv[j] = ok ? w : NONE; });
#pragma unroll
    for (int g = 0; g < KW_SEG; g += BS_G) {
        uint32_t b[BS_G];
        buckets_of<W, BS_G>(spl, ct2, mp, wv + g, b);
#pragma unroll
        for (int j = 0; j < BS_G; j++)
            if (wv[g + j] != NONE) atomicAdd(&h[b[j]], 1u);
    }
    __syncthreads();
    uint32_t c[BS_NB / BT_THREADS], o[BS_NB / BT_THREADS];
#pragma unroll
    for (int e = 0; e < (int)(BS_NB / BT_THREADS); e++) c[e] = h[e * BT_THREADS + threadIdx.x];
#pragma unroll
    for (int e = 0; e < (int)(BS_NB / BT_THREADS); e++)
        if (c[e]) o[e] = atomicAdd(&cnt[e * BT_THREADS + threadIdx.x], c[e]);
#pragma unroll
    for (int e = 0; e < (int)(BS_NB / BT_THREADS); e++)
        if (c[e]) wgoff[(uint64_t)tile * BS_NB + e * BT_THREADS + threadIdx.x] = o[e];
}
template <int K>
__global__ __launch_bounds__(BsGeo<typename Windows<K>::Word>::THREADS) void bs_hist_kernel(
    const uint8_t *__restrict__ clean, uint64_t len, typename Windows<K>::Word lo, typename Windows<K>::Word hi,
    const typename Windows<K>::Word *__restrict__ spl_g, const uint16_t *__restrict__ ct_g, const BsMap<typename Windows<K>::Word> mp,
    uint32_t *__restrict__ cnt, uint32_t *__restrict__ wgoff)
{
    bs_hist_body<K>(clean, len, lo, hi, spl_g, ct_g, mp, cnt, wgoff, blockIdx.x);
}
template <int K>
__global__ __launch_bounds__(BsGeo<typename Windows<K>::Word>::THREADS) void bs_hist_batch_kernel(const BsBatch<typename Windows<K>::Word> p)
{
    const BsItem<typename Windows<K>::Word> &it = p.it[bs_batch_sample(p, blockIdx.x)];
    bs_hist_body<K>(it.clean, it.len, p.lo, p.hi, p.spl, p.ct, p.mp, it.cnt, it.wgoff, blockIdx.x - it.tile0);
}

template <int K>
__device__ __forceinline__ void bs_partition_body(const uint8_t *__restrict__ clean, uint64_t len, typename Windows<K>::Word lo,
                                                  typename Windows<K>::Word hi, const typename Windows<K>::Word *__restrict__ spl_g,
                                                  const uint16_t *__restrict__ ct_g, const BsMap<typename Windows<K>::Word> mp,
                                                  const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ wgoff,
                                                  uint32_t *__restrict__ base_out, typename Windows<K>::Word *__restrict__ part,
                                                  const uint32_t tile)
{
    typedef typename Windows<K>::Word W;
    constexpr int BT_THREADS = BsGeo<W>::THREADS, BT_TILE = BsGeo<W>::TILE;
    constexpr int BPT = BS_NB / BT_THREADS;               // buckets a thread owns in the offsets step: 4 (32-bit words) or 8
    extern __shared__ uint64_t dyn_lds64[];               // bp_lds_bytes<W>()
    W *stage = reinterpret_cast<W *>(dyn_lds64);          // 64 KB: the tile's words, bucketed
    W *spl = stage + BT_TILE;
    uint32_t *h = reinterpret_cast<uint32_t *>(spl + BS_NB);   // counts, then (start of this tile's range in the bucket) - (local start)
    uint16_t *lstart = reinterpret_cast<uint16_t *>(h + BS_NB);
    uint32_t *scan_lds = h + BS_NB + BS_NB / 2;
    uint32_t *ct2 = scan_lds + 16;
    uint16_t *stageb = reinterpret_cast<uint16_t *>(scan_lds + 16 + BS_CELLS);   // the bucket of every staged word
    constexpr W NONE = (W)~(W)0;                          // (a valid canonical word is never all ones: its reverse complement would be 0)
    for (uint32_t d = threadIdx.x; d < BS_NB; d += BT_THREADS) h[d] = 0;
    load_map2<W>(spl, ct2, spl_g, ct_g, BT_THREADS);
    __syncthreads();
    const uint64_t s = ((uint64_t)tile * BT_THREADS + threadIdx.x) * KW_SEG;
    typename Windows<K>::St st;
    load_streams(st, clean, len, s);
    W wv[KW_SEG];          // the word
    uint32_t rb[KW_SEG];   // bucket << 16 | rank inside (tile, bucket)
    Windows<K>::run(st, lo, hi, [&](int j, bool ok, W w) { wv[j] = ok ? w : NONE; });
#pragma unroll
    for (int g = 0; g < KW_SEG; g += BS_G) {
        uint32_t b[BS_G];
        buckets_of<W, BS_G>(spl, ct2, mp, wv + g, b);
        // (the ranks: G returning atomics in flight, their results waited for together)
#pragma unroll
        for (int j = 0; j < BS_G; j++) rb[g + j] = wv[g + j] != NONE ? atomicAdd(&h[b[j]], 1u) : 0u;
#pragma unroll
        for (int j = 0; j < BS_G; j++) rb[g + j] |= b[j] << 16;
    }
    __syncthreads();
    // thread t owns buckets BPT t .. BPT t + BPT - 1: local starts, global bucket bases, this tile's offset in each bucket
    uint32_t ltot;
    {
        uint32_t c4[BPT], g4[BPT], lsum = 0, gsum = 0;
        const uint32_t d0 = threadIdx.x * BPT;
#pragma unroll
        for (int e = 0; e < BPT; e++) {
            c4[e] = h[d0 + e];
            g4[e] = cnt[d0 + e];
            lsum += c4[e];
            gsum += g4[e];
        }
        uint32_t gtot;
        uint32_t lex = psk_block_excl_scan_u32<BT_THREADS>(lsum, &ltot, scan_lds);
        uint32_t gex = psk_block_excl_scan_u32<BT_THREADS>(gsum, &gtot, scan_lds);
#pragma unroll
        for (int e = 0; e < BPT; e++) {
            const uint32_t d = d0 + e;
            lstart[d] = (uint16_t)lex;
            h[d] = gex + (c4[e] ? wgoff[(uint64_t)tile * BS_NB + d] : 0u) - lex;
            if (tile == 0) base_out[d] = gex;
            lex += c4[e];
            gex += g4[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KW_SEG; j++)
        if (wv[j] != NONE) {
            const uint32_t at = (uint32_t)lstart[rb[j] >> 16] + (rb[j] & 0xffffu);
            stage[at] = wv[j];
            stageb[at] = (uint16_t)(rb[j] >> 16);
        }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ltot; i += BT_THREADS) part[(size_t)(uint32_t)(h[stageb[i]] + i)] = stage[i];
}
template <int K>
__global__ __launch_bounds__(BsGeo<typename Windows<K>::Word>::THREADS) void bs_partition_kernel(
    const uint8_t *__restrict__ clean, uint64_t len, typename Windows<K>::Word lo, typename Windows<K>::Word hi,
    const typename Windows<K>::Word *__restrict__ spl_g, const uint16_t *__restrict__ ct_g, const BsMap<typename Windows<K>::Word> mp,
    const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ wgoff, uint32_t *__restrict__ base_out,
    typename Windows<K>::Word *__restrict__ part)
{
    bs_partition_body<K>(clean, len, lo, hi, spl_g, ct_g, mp, cnt, wgoff, base_out, part, blockIdx.x);
}
template <int K>
__global__ __launch_bounds__(BsGeo<typename Windows<K>::Word>::THREADS) void bs_partition_batch_kernel(const BsBatch<typename Windows<K>::Word> p)
{
    const BsItem<typename Windows<K>::Word> &it = p.it[bs_batch_sample(p, blockIdx.x)];
    bs_partition_body<K>(it.clean, it.len, p.lo, p.hi, p.spl, p.ct, p.mp, it.cnt, it.wgoff, it.base, it.part, blockIdx.x - it.tile0);
}

// One workgroup per bucket: sort, run lengths.  The words of a bucket spread evenly over its narrow range (the buckets
// have equal counts), so they are dealt into 2,048 sub-bins by the top bits of (word - first word of the bucket) -- one
// or two words per sub-bin -- and a thread straightens the stretch of its consecutive sub-bins in registers: a handful
// of barriers instead of the 78 of a bitonic network over 4,096 keys (33 us per bucket against 15).  A sub-bin with more than BS_BIN_MAX words (the
// sample is unlike the one the splitters came from) sends the sample to the fall-back, like a bucket beyond `cap`.
// wtmp / ctmp: unique words and their counts, written from base[b] on (a bucket has at most cnt[b] of them and the
// buckets' key ranges lie back to back).
constexpr uint32_t BS_BINS = 2048;
constexpr uint32_t BS_BIN_MAX = 192;
template <typename W>
__device__ __forceinline__ void bs_sort_bucket(const uint32_t b, const W *__restrict__ part, const uint32_t *__restrict__ cnt,
                                               const uint32_t *__restrict__ base, const W *__restrict__ spl,
                                               uint32_t nb, W last_word, uint32_t cap, W *__restrict__ wtmp,
                                               uint32_t *__restrict__ ctmp, uint32_t *__restrict__ uniq_out,
                                               uint32_t *__restrict__ flag, uint64_t *dyn_lds)
{
    constexpr int T = SortGeo<W>::THREADS;
    constexpr int BPT = BS_BINS / T;        // sub-bins per thread
    constexpr int KEEP = 6;                 // words a thread holds in registers between the counting and the dealing pass (a bucket of
                                            // <= KEEP x T words -- every ordinary one -- is read from global memory once)
    constexpr W NONE = (W)~(W)0;
    W *key = reinterpret_cast<W *>(dyn_lds);   // the bucket's words dealt into sub-bins, then sorted
    uint32_t *start = reinterpret_cast<uint32_t *>(key + BS_CAP_MAX);     // sub-bin counts -> starts
    uint32_t *fill = start + BS_BINS;
    uint32_t *scan_lds = fill + BS_BINS;
    uint16_t *pos = reinterpret_cast<uint16_t *>(start);   // after the sort: where the run heads are (start and fill are done with by then)
    const uint32_t t = threadIdx.x;
#ifdef PSK_BS_STAMPS
    long long *stamps = reinterpret_cast<long long *>(flag + 2);
    if (b == 100 && t == 0) stamps[6] = clock64();
#endif
    const uint32_t n = cnt[b];
    if (n == 0) { if (t == 0) uniq_out[b] = 0; return; }
    if (n > cap) {   // not this way: the host sends the sample through the radix sort (bucket_chain_finalize)
        if (t == 0) { uniq_out[b] = 0; atomicOr(flag, 1u); }
        return;
    }
    const size_t off = base[b];
    const W first = spl[b], last = b + 1 < nb ? (W)(spl[b + 1] - 1u) : last_word;   // the bucket's words lie in [first, last]
    const W span = last - first;
    const uint32_t sh = span < (W)BS_BINS ? 0u : (64u - (uint32_t)__builtin_clzll((unsigned long long)span)) - 11u;   // (w - first) >> sh < 2048
    W held[KEEP];
#pragma unroll
    for (int q = 0; q < KEEP; q++) held[q] = t + q * T < n ? part[off + t + q * T] : NONE;
    for (uint32_t i = t; i < BS_BINS; i += T) start[i] = 0;
    __syncthreads();
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[0] = clock64();
#endif
#pragma unroll
    for (int q = 0; q < KEEP; q++)
        if (t + q * T < n) atomicAdd(&start[(uint32_t)((held[q] - first) >> sh)], 1u);
    for (uint32_t i = t + KEEP * T; i < n; i += T) atomicAdd(&start[(uint32_t)((part[off + i] - first) >> sh)], 1u);
    __syncthreads();
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[1] = clock64();
#endif
    {
        uint32_t c4[BPT], sum = 0, big = 0;
#pragma unroll
        for (int e = 0; e < BPT; e++) { c4[e] = start[BPT * t + e]; sum += c4[e]; big |= c4[e] > BS_BIN_MAX; }
        // one scan for both: the sub-bin counts in the low bits, "a sub-bin of mine is too full" in bit 24 and up
        uint32_t tot;
        uint32_t ex = psk_block_excl_scan_u32<T>(sum | (big << 24), &tot, scan_lds) & 0xffffffu;
        if (tot >> 24) {   // uniform: every thread has the same total
            if (t == 0) { uniq_out[b] = 0; atomicOr(flag, 1u); }
            return;
        }
#pragma unroll
        for (int e = 0; e < BPT; e++) { start[BPT * t + e] = ex; fill[BPT * t + e] = ex; ex += c4[e]; }
    }
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[2] = clock64();
#endif
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KEEP; q++)
        if (t + q * T < n) key[atomicAdd(&fill[(uint32_t)((held[q] - first) >> sh)], 1u)] = held[q];
    for (uint32_t i = t + KEEP * T; i < n; i += T) {
        const W w = part[off + i];
        key[atomicAdd(&fill[(uint32_t)((w - first) >> sh)], 1u)] = w;
    }
    __syncthreads();
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[3] = clock64();
#endif
    // thread t straightens its BPT consecutive sub-bins as ONE stretch (every word of a sub-bin is below every word of the
    // next): up to 16 words in registers through Batcher's odd-even merge network (63 compare-exchanges, no LDS round
    // trips -- insertion inside LDS, a few dependent reads per step with the lanes of a wave waiting for the longest
    // sub-bin, was 60 % of the kernel); longer stretches (rare) by insertion in LDS
    {
        const uint32_t lo = start[BPT * t], hi = fill[BPT * t + BPT - 1], m = hi - lo;
        if (m > 16) {
            for (uint32_t i = lo + 1; i < hi; i++) {
                const W x = key[i];
                uint32_t j = i;
                while (j > lo && key[j - 1] > x) { key[j] = key[j - 1]; j--; }
                key[j] = x;
            }
        } else if (m > 1) {
            W r[16];
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = (uint32_t)i < m ? key[lo + i] : NONE;
#pragma unroll
            for (int p = 1; p < 16; p *= 2)
#pragma unroll
                for (int k = p; k >= 1; k /= 2)
#pragma unroll
                    for (int j = k % p; j <= 16 - 1 - k; j += 2 * k)
#pragma unroll
                        for (int i = 0; i <= (k - 1 < 16 - j - k - 1 ? k - 1 : 16 - j - k - 1); i++)
                            if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                                const W x = r[i + j], y = r[i + j + k];
                                r[i + j] = x < y ? x : y;
                                r[i + j + k] = x < y ? y : x;
                            }
#pragma unroll
            for (int i = 0; i < 16; i++)
                if ((uint32_t)i < m) key[lo + i] = r[i];
        }
    }
    __syncthreads();
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[4] = clock64();
#endif
    // run lengths: thread t owns the sorted words [t c, (t + 1) c); it counts the run heads in its stretch, one scan gives
    // their ranks, and every head writes its word and its position (pos) -- the length of run r is then the difference of
    // the positions of heads r + 1 and r
    const uint32_t c = (n + T - 1) / T;
    const uint32_t i0 = t * c < n ? t * c : n, i1 = i0 + c < n ? i0 + c : n;
    uint32_t heads = 0;
    for (uint32_t i = i0; i < i1; i++) heads += (i == 0 || key[i - 1] != key[i]) ? 1u : 0u;
    uint32_t nu;
    uint32_t r = psk_block_excl_scan_u32<T>(heads, &nu, scan_lds);
    for (uint32_t i = i0; i < i1; i++)
        if (i == 0 || key[i - 1] != key[i]) {
            pos[r] = (uint16_t)i;
            wtmp[off + r] = key[i];
            r++;
        }
    __syncthreads();
    for (uint32_t q = t; q < nu; q += T) ctmp[off + q] = (q + 1 < nu ? (uint32_t)pos[q + 1] : n) - (uint32_t)pos[q];
#ifdef PSK_BS_STAMPS
    if (b == 100 && t == 0) stamps[5] = clock64();
#endif
    if (t == 0) uniq_out[b] = nu;
}

// a workgroup takes buckets blockIdx.x, blockIdx.x + gridDim.x, ...: one resident set of workgroups for the whole sample (two
// per CU) instead of four launch rounds of them
template <typename W>
__global__ __launch_bounds__(SortGeo<W>::THREADS, SortGeo<W>::WAVES_PER_SIMD) void bs_sort_kernel(const W *__restrict__ part, const uint32_t *__restrict__ cnt,
                                                                   const uint32_t *__restrict__ base, const W *__restrict__ spl,
                                                                   uint32_t nb, W last_word, uint32_t cap, W *__restrict__ wtmp,
                                                                   uint32_t *__restrict__ ctmp, uint32_t *__restrict__ uniq_out,
                                                                   uint32_t *__restrict__ flag)
{
    extern __shared__ uint64_t dyn_lds64[];   // bsort_lds_bytes<W>()
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        bs_sort_bucket<W>(b, part, cnt, base, spl, nb, last_word, cap, wtmp, ctmp, uniq_out, flag, dyn_lds64);
        __syncthreads();   // the next bucket reuses the arrays
    }
}

template <typename W>
__global__ __launch_bounds__(SortGeo<W>::THREADS, SortGeo<W>::WAVES_PER_SIMD) void bs_sort_batch_kernel(const BsBatch<W> p)
{
    extern __shared__ uint64_t dyn_lds64[];   // bsort_lds_bytes<W>()
    const BsItem<W> &it = p.it[blockIdx.y];
    for (uint32_t b = blockIdx.x; b < p.mp.nb; b += gridDim.x) {
        bs_sort_bucket<W>(b, it.part, it.cnt, it.base, p.spl, p.mp.nb, p.last_word, p.cap, it.wtmp, it.ctmp, it.uniq, it.flag, dyn_lds64);
        __syncthreads();
    }
}

// uniq -> offsets of the buckets in the list; totals: host[0] = words kept, [1] = unique words, [3] = a bucket overflowed
__device__ __forceinline__ void bs_totals_body(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ uniq,
                                               uint32_t *__restrict__ uoff, const uint32_t *__restrict__ flag, uint32_t nb,
                                               uint32_t *__restrict__ host)
{
    __shared__ uint32_t scan_lds[16];
    const uint32_t t = threadIdx.x;
    const uint32_t u0 = 2 * t < nb ? uniq[2 * t] : 0u, u1 = 2 * t + 1 < nb ? uniq[2 * t + 1] : 0u, c = cnt[2 * t] + cnt[2 * t + 1];
    uint32_t utot, ctot;
    const uint32_t ex = psk_block_excl_scan_u32<1024>(u0 + u1, &utot, scan_lds);
    psk_block_excl_scan_u32<1024>(c, &ctot, scan_lds);
    uoff[2 * t] = ex;
    uoff[2 * t + 1] = ex + u0;
    if (t == 0) { host[0] = ctot; host[1] = utot; host[2] = 0; host[3] = *flag; }
}

__global__ __launch_bounds__(1024) void bs_totals_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ uniq,
                                                         uint32_t *__restrict__ uoff, const uint32_t *__restrict__ flag,
                                                         uint32_t nb, uint32_t *__restrict__ host)
{
    bs_totals_body(cnt, uniq, uoff, flag, nb, host);
}
template <typename W>
__global__ __launch_bounds__(1024) void bs_totals_batch_kernel(const BsBatch<W> p)
{
    const BsItem<W> &it = p.it[blockIdx.x];
    bs_totals_body(it.cnt, it.uniq, it.uoff, it.flag, p.mp.nb, it.host);
}

// (one flat pass with a search over the offsets per element was slower: 41 us against 26 for a whole genome)
template <typename W>
__device__ __forceinline__ void bs_compact_body(const W *__restrict__ wtmp, const uint32_t *__restrict__ ctmp,
                                                const uint32_t *__restrict__ base, const uint32_t *__restrict__ uniq,
                                                const uint32_t *__restrict__ uoff, uint64_t *__restrict__ words,
                                                uint32_t *__restrict__ freqs, const uint32_t bucket)
{
    const uint32_t b = bucket, n = uniq[b];
    const size_t src = base[b], dst = uoff[b];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        words[dst + i] = wtmp[src + i];
        freqs[dst + i] = ctmp[src + i];
    }
}
template <typename W>
__global__ void bs_compact_kernel(const W *__restrict__ wtmp, const uint32_t *__restrict__ ctmp, const uint32_t *__restrict__ base,
                                  const uint32_t *__restrict__ uniq, const uint32_t *__restrict__ uoff, uint64_t *__restrict__ words,
                                  uint32_t *__restrict__ freqs)
{
    bs_compact_body<W>(wtmp, ctmp, base, uniq, uoff, words, freqs, blockIdx.x);
}
template <typename W>
__global__ void bs_compact_batch_kernel(const BsBatch<W> p)
{
    const BsItem<W> &it = p.it[blockIdx.y];
    bs_compact_body<W>(it.wtmp, it.ctmp, it.base, it.uniq, it.uoff, it.words, it.freqs, blockIdx.x);
}

// the partitioned words of a sample as u64 keys for the radix sort (the fall-back)
template <typename W>
__global__ void bs_expand_kernel(const W *__restrict__ part, uint64_t n, uint64_t *__restrict__ keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = part[i];
}

struct BsBufs {
    uint32_t *cnt, *flag, *base, *uniq, *uoff;
};

BsBufs bs_bufs(const CountLane &L, uint32_t slot)
{
    BsBufs d;
    d.cnt = L.dc_cnt.as<uint32_t>() + (size_t)slot * (BS_NB + 16);
    d.flag = d.cnt + BS_NB;
    d.base = L.dc_meta.as<uint32_t>();
    d.uniq = d.base + BS_NB;
    d.uoff = d.uniq + BS_NB;
    return d;
}

uint32_t bs_cap()
{
    const char *e = getenv("PSK_BS_CAP");   // tests: a small capacity forces the fall-back (read per call)
    const long v = e ? atol(e) : 0;
    return (uint32_t)(v > 0 && v < (long)BS_CAP_MAX ? v : (long)BS_CAP_MAX);
}

template <int K>
int launch_tiles(psk_ctx *ctx, CountLane &L, const BsBufs &d, uint64_t clean_len, uint32_t n_tiles, typename Windows<K>::Word lo,
                 typename Windows<K>::Word hi)
{
    typedef typename Windows<K>::Word W;
    constexpr int BT_THREADS = BsGeo<W>::THREADS;
    const BsMap<W> mp{(W)ctx->bs_lo, ctx->bs_shift, ctx->bs_nb};
    const uint16_t *ct = ctx->bs_ct.as<uint16_t>();
    bs_hist_kernel<K><<<n_tiles, BT_THREADS, 0, ctx->stream>>>(L.raw.as<uint8_t>(), clean_len, lo, hi, ctx->bs_spl.as<W>(), ct, mp, d.cnt,
                                                               L.dc_wgoff.as<uint32_t>());
    PSK_HIP(ctx, hipGetLastError());
    static PerDeviceOnce attr_set;
    if (attr_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bs_partition_kernel<K>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_bytes<W>()));
    }
    bs_partition_kernel<K><<<n_tiles, BT_THREADS, bp_lds_bytes<W>(), ctx->stream>>>(L.raw.as<uint8_t>(), clean_len, lo, hi,
                                                                                  ctx->bs_spl.as<W>(), ct, mp, d.cnt,
                                                                                  L.dc_wgoff.as<uint32_t>(), d.base, L.dc_part.as<W>());
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

// k -> the instantiation: F<K>::run(args...) for the K of width W (14..16 | 17..32)
template <typename W, template <int> class F, typename... A>
int for_k(psk_ctx *ctx, A &&...a)
{
    switch (ctx->k) {
#define PSK_BS_CASE(K_) case K_: if constexpr (sizeof(typename Windows<K_>::Word) == sizeof(W)) return F<K_>::run(ctx, a...); else break;
    PSK_BS_CASE(14) PSK_BS_CASE(15) PSK_BS_CASE(16) PSK_BS_CASE(17) PSK_BS_CASE(18) PSK_BS_CASE(19) PSK_BS_CASE(20) PSK_BS_CASE(21)
    PSK_BS_CASE(22) PSK_BS_CASE(23) PSK_BS_CASE(24) PSK_BS_CASE(25) PSK_BS_CASE(26) PSK_BS_CASE(27) PSK_BS_CASE(28) PSK_BS_CASE(29)
    PSK_BS_CASE(30) PSK_BS_CASE(31) PSK_BS_CASE(32)
#undef PSK_BS_CASE
    default: break;
    }
    return psk_fail(ctx, PSK_ESTATE, "the bucketed sort of %d-byte words is not built for k = %d", (int)sizeof(W), ctx->k);
}

// the slab's bounds as words of the run's width: [lo, hi), hi = the first word beyond the slab or the space (all ones where that
// is 2^32 / 2^64: no canonical word is all ones)
template <typename W>
void slab_bounds_w(const psk_ctx *ctx, W *lo, W *hi)
{
    const W none = (W)~(W)0;
    const int bits = 2 * ctx->k;
    const bool whole = bits >= (int)(8 * sizeof(W));                    // the space fills the word: k = 16 / k = 32
    const uint64_t space = bits >= 64 ? 0 : (1ull << bits);
    *lo = (W)ctx->slab_lo;
    if (ctx->slab_hi && (bits >= 64 || ctx->slab_hi < space) && ctx->slab_hi <= (uint64_t)none) *hi = (W)ctx->slab_hi;
    else *hi = whole ? none : (W)space;
}

template <typename W>
int chain_enqueue_w(psk_ctx *ctx, CountLane &L, uint64_t clean_len, uint64_t n);

template <int K>
struct ChainTiles {
    static int run(psk_ctx *ctx, CountLane &L, const BsBufs &d, uint64_t clean_len, uint32_t n_tiles)
    {
        typename Windows<K>::Word lo, hi;
        slab_bounds_w(ctx, &lo, &hi);
        return launch_tiles<K>(ctx, L, d, clean_len, n_tiles, lo, hi);
    }
};

template <typename W>
int chain_enqueue_w(psk_ctx *ctx, CountLane &L, uint64_t clean_len, uint64_t n)
{
    W lo, hi;
    slab_bounds_w(ctx, &lo, &hi);
    const uint32_t n_tiles = div_up(clean_len, BsGeo<W>::TILE);
    PSK_TRY(dev_reserve(ctx, L.dc_part, n * sizeof(W) + 64));
    PSK_TRY(dev_reserve(ctx, L.dc_wgoff, (size_t)n_tiles * BS_NB * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_cnt, (size_t)BS_SLOTS * (BS_NB + 16) * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_meta, (size_t)(4 * BS_NB + 4) * 4));
    PSK_TRY(dev_reserve(ctx, L.dc_mtemp, (n + 8) * (sizeof(W) + 4) + 64));
    if (L.dc_slot == 0 || L.dc_slot >= BS_SLOTS) {
        PSK_HIP(ctx, hipMemsetAsync(L.dc_cnt.p, 0, (size_t)BS_SLOTS * (BS_NB + 16) * 4, ctx->stream));
        L.dc_slot = 0;
    }
    const BsBufs d = bs_bufs(L, L.dc_slot++);
    PSK_TRY((for_k<W, ChainTiles>(ctx, L, d, clean_len, n_tiles)));
    PSK_HIP(ctx, hipEventRecord(L.raw_free, ctx->stream));
    L.raw_used = true;
    static PerDeviceOnce attr_set;
    if (attr_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bs_sort_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)bsort_lds_bytes<W>()));
    }
    W *wtmp = L.dc_mtemp.as<W>();
    uint32_t *ctmp = reinterpret_cast<uint32_t *>(wtmp + n + 8);
    const uint32_t resident = (uint32_t)SortGeo<W>::WGS_PER_CU * (uint32_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    const uint32_t sort_wgs = ctx->bs_nb < resident ? ctx->bs_nb : resident;
    bs_sort_kernel<W><<<sort_wgs, SortGeo<W>::THREADS, bsort_lds_bytes<W>(), ctx->stream>>>(L.dc_part.as<W>(), d.cnt, d.base, ctx->bs_spl.as<W>(),
                                                                                          ctx->bs_nb, (W)(hi - 1u), bs_cap(), wtmp, ctmp, d.uniq, d.flag);
    PSK_HIP(ctx, hipGetLastError());
    bs_totals_kernel<<<1, 1024, 0, ctx->stream>>>(d.cnt, d.uniq, d.uoff, d.flag, ctx->bs_nb, L.pinned_cnt);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipEventRecord(L.done, ctx->stream));
    L.bs = true;
    return PSK_OK;
}

template <typename W>
int splitters_from_w(psk_ctx *ctx, const SampleList &S, uint32_t nb)
{
    W lo, hi;
    slab_bounds_w(ctx, &lo, &hi);
    // cells over [lo, hi): (hi - lo - 1) >> shift < BS_CELLS   (hi = all ones stands for one more: the difference is a cell at most)
    uint32_t shift = 0;
    while (((uint64_t)(W)(hi - lo - 1u) >> shift) >= (uint64_t)BS_CELLS) shift++;
    ctx->bs_lo = (uint64_t)lo;
    ctx->bs_shift = shift;
    PSK_TRY(dev_reserve(ctx, ctx->bs_spl, (size_t)BS_NB * sizeof(W)));
    PSK_TRY(dev_reserve(ctx, ctx->bs_ct, (size_t)(BS_CELLS + 2) * 2));
    bs_splitters_kernel<W><<<div_up(BS_NB, 256), 256, 0, ctx->stream>>>(S.words, S.n_unique, nb, lo, ctx->bs_spl.as<W>());
    PSK_HIP(ctx, hipGetLastError());
    bs_cells_kernel<W><<<div_up(BS_CELLS + 1, 256), 256, 0, ctx->stream>>>(ctx->bs_spl.as<W>(), BsMap<W>{lo, shift, nb}, ctx->bs_ct.as<uint16_t>());
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

template <typename W>
int chain_finalize_w(psk_ctx *ctx, CountLane &L, SampleList &S, const BsBufs &d)
{
    const W *wtmp = L.dc_mtemp.as<W>();
    const uint32_t *ctmp = reinterpret_cast<const uint32_t *>(wtmp + L.n + 8);
    bs_compact_kernel<W><<<ctx->bs_nb, 512, 0, ctx->stream>>>(wtmp, ctmp, d.base, d.uniq, d.uoff, S.words, S.freqs);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

template <int K>
struct GroupTiles {
    static int run(psk_ctx *ctx, const BsBatch<typename Windows<K>::Word> &p, uint32_t tiles)
    {
        typedef typename Windows<K>::Word W;
        constexpr int BT_THREADS = BsGeo<W>::THREADS;
        bs_hist_batch_kernel<K><<<tiles, BT_THREADS, 0, ctx->stream>>>(p);
        PSK_HIP(ctx, hipGetLastError());
        static PerDeviceOnce attr_set;
        if (attr_set.first(ctx->device)) {
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bs_partition_batch_kernel<K>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp_lds_bytes<W>()));
        }
        bs_partition_batch_kernel<K><<<tiles, BT_THREADS, bp_lds_bytes<W>(), ctx->stream>>>(p);
        PSK_HIP(ctx, hipGetLastError());
        return PSK_OK;
    }
};

template <typename W>
int group_enqueue_w(psk_ctx *ctx, CountLane *const *lanes, const uint64_t *clean_len, const uint64_t *n, int count)
{
    BsBatch<W> p;
    memset(&p, 0, sizeof(p));
    p.n = (uint32_t)count;
    slab_bounds_w(ctx, &p.lo, &p.hi);
    p.last_word = (W)(p.hi - 1u);
    p.cap = bs_cap();
    p.mp = BsMap<W>{(W)ctx->bs_lo, ctx->bs_shift, ctx->bs_nb};
    p.spl = ctx->bs_spl.as<W>();
    p.ct = ctx->bs_ct.as<uint16_t>();
    uint32_t tiles = 0;
    for (int s = 0; s < count; s++) {
        CountLane &L = *lanes[s];
        const uint32_t n_tiles = div_up(clean_len[s], BsGeo<W>::TILE);
        PSK_TRY(dev_reserve(ctx, L.dc_part, n[s] * sizeof(W) + 64));
        PSK_TRY(dev_reserve(ctx, L.dc_wgoff, (size_t)n_tiles * BS_NB * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_cnt, (size_t)BS_SLOTS * (BS_NB + 16) * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_meta, (size_t)(4 * BS_NB + 4) * 4));
        PSK_TRY(dev_reserve(ctx, L.dc_mtemp, (n[s] + 8) * (sizeof(W) + 4) + 64));
        if (L.dc_slot == 0 || L.dc_slot >= BS_SLOTS) {
            PSK_HIP(ctx, hipMemsetAsync(L.dc_cnt.p, 0, (size_t)BS_SLOTS * (BS_NB + 16) * 4, ctx->stream));
            L.dc_slot = 0;
        }
        const BsBufs d = bs_bufs(L, L.dc_slot++);
        BsItem<W> &it = p.it[s];
        it.clean = L.raw.as<uint8_t>();
        it.len = clean_len[s];
        it.cnt = d.cnt; it.wgoff = L.dc_wgoff.as<uint32_t>(); it.base = d.base; it.part = L.dc_part.as<W>();
        it.wtmp = L.dc_mtemp.as<W>(); it.ctmp = reinterpret_cast<uint32_t *>(it.wtmp + n[s] + 8);
        it.uniq = d.uniq; it.uoff = d.uoff; it.flag = d.flag; it.host = L.pinned_cnt;
        it.tile0 = tiles;
        tiles += n_tiles;
    }
    PSK_TRY((for_k<W, GroupTiles>(ctx, p, tiles)));
    for (int s = 0; s < count; s++) {
        PSK_HIP(ctx, hipEventRecord(lanes[s]->raw_free, ctx->stream));
        lanes[s]->raw_used = true;
    }
    static PerDeviceOnce attr_set;
    if (attr_set.first(ctx->device)) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bs_sort_batch_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)bsort_lds_bytes<W>()));
    }
    // one resident set of workgroups for the whole group (two per CU by LDS; one with 64-bit words)
    const uint32_t wg_all = (uint32_t)SortGeo<W>::WGS_PER_CU * (uint32_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    uint32_t per = (wg_all + (uint32_t)count - 1) / (uint32_t)count;
    if (per > ctx->bs_nb) per = ctx->bs_nb;
    if (per < 1) per = 1;
    bs_sort_batch_kernel<W><<<dim3(per, (uint32_t)count), SortGeo<W>::THREADS, bsort_lds_bytes<W>(), ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    bs_totals_batch_kernel<W><<<(uint32_t)count, 1024, 0, ctx->stream>>>(p);
    PSK_HIP(ctx, hipGetLastError());
    for (int s = 0; s < count; s++) {
        PSK_HIP(ctx, hipEventRecord(lanes[s]->done, ctx->stream));
        lanes[s]->bs = true;
        lanes[s]->dc_defer_compact = true;
    }
    return PSK_OK;
}

template <typename W>
int group_compact_w(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, int count)
{
    BsBatch<W> p;
    memset(&p, 0, sizeof(p));
    uint32_t m = 0;
    for (int s = 0; s < count; s++) {
        CountLane &L = *lanes[s];
        L.dc_defer_compact = false;
        const SampleList &S = ctx->lists[sample_idx[s]];
        if (!S.n_unique || !S.words) continue;
        const BsBufs d = bs_bufs(L, L.dc_slot - 1);
        BsItem<W> &it = p.it[m++];
        it.wtmp = L.dc_mtemp.as<W>(); it.ctmp = reinterpret_cast<uint32_t *>(it.wtmp + L.n + 8);
        it.base = d.base; it.uniq = d.uniq; it.uoff = d.uoff;
        it.words = S.words; it.freqs = S.freqs;
    }
    p.n = m;
    if (m) {
        bs_compact_batch_kernel<W><<<dim3(ctx->bs_nb, m), 512, 0, ctx->stream>>>(p);
        PSK_HIP(ctx, hipGetLastError());
    }
    return PSK_OK;
}

inline bool bs_wide(const psk_ctx *ctx) { return ctx->k > 16; }
inline bool bs_k_ok(const psk_ctx *ctx) { return ctx->k >= 14 && ctx->k <= 32; }

}  // namespace

// n = the sample's windows (an upper bound of its words under a slab filter: bs_keep is the share the run's first list kept).
// A sample that would put more than half the LDS sort's capacity into an average bucket (reads at depth) keeps the radix
// route: its buckets would overflow and the work would be done twice.
bool bucket_route_ok(const psk_ctx *ctx, uint64_t n)
{
    if (!ctx->bs_ready || ctx->dense_mode || !bs_k_ok(ctx)) return false;
    return (double)n * ctx->bs_keep / ctx->bs_nb <= 0.5 * BS_CAP_MAX;
}

// after a sample of the run has gone through the radix route: its list gives the splitters of the later ones
int bucket_splitters_from(psk_ctx *ctx, const SampleList &S, uint64_t windows)
{
    if (ctx->bs_ready || ctx->dense_mode || !bs_k_ok(ctx) || S.n_unique < 32768 || !S.words) return PSK_OK;
    if (getenv("PSK_NO_BUCKET_SORT")) return PSK_OK;   // read per call: tests cross the two routes in one process
    // about 2,400 words per bucket (the LDS sort takes 8,192): a genome gives 2,048 buckets, a rank's eighth of it 256
    uint32_t nb = 64;
    while (nb < BS_NB && (uint64_t)nb * 2400 < S.n_unique) nb <<= 1;
    ctx->bs_nb = nb;
    ctx->bs_keep = windows ? (double)S.n_total / (double)windows : 1.0;
    if (ctx->bs_keep > 1.0) ctx->bs_keep = 1.0;
    PSK_TRY(bs_wide(ctx) ? splitters_from_w<uint64_t>(ctx, S, nb) : splitters_from_w<uint32_t>(ctx, S, nb));
    ctx->bs_ready = true;
    return PSK_OK;
}

int bucket_chain_enqueue(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n)
{
    (void)sample_idx;
    return bs_wide(ctx) ? chain_enqueue_w<uint64_t>(ctx, L, clean_len, n) : chain_enqueue_w<uint32_t>(ctx, L, clean_len, n);
}

// after hipEventSynchronize(L.done): the arena block and the packing pass; or, when a bucket overflowed, the radix sort of
// the partitioned words (they are all there, bucket by bucket) and the ordinary run-length passes
int bucket_chain_finalize(psk_ctx *ctx, CountLane &L, SampleList &S, uint64_t n_kept, uint64_t nu, bool *fell_back)
{
    const BsBufs d = bs_bufs(L, L.dc_slot - 1);
    L.bs = false;
    *fell_back = false;
#ifdef PSK_BS_STAMPS
    {
        long long st[7];
        (void)hipMemcpy(st, d.flag + 2, sizeof(st), hipMemcpyDeviceToHost);
        fprintf(stderr, "bs_sort bucket 100: prologue %lld  zero %lld  deal-count %lld  scan %lld  deal %lld  sort %lld  runs %lld  (cycles)\n",
                st[0] - st[6], 0ll, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
    }
#endif
    if (L.pinned_cnt[3]) {
        if (getenv("PSK_TRACE")) fprintf(stderr, "bucketed sort: a sample takes the fall-back (a bucket or a sub-bin overflowed)\n");
        *fell_back = true;
        return PSK_OK;   // the caller runs the radix route on dc_part (bucket_fallback_keys)
    }
    if (!nu) return PSK_OK;
    PSK_TRY(arena_alloc(ctx, nu * 8, (void **)&S.words));
    PSK_TRY(arena_alloc(ctx, nu * 4, (void **)&S.freqs));
    (void)n_kept;
    if (L.dc_defer_compact) return PSK_OK;   // the group packs in one launch (bucket_group_compact)
    return bs_wide(ctx) ? chain_finalize_w<uint64_t>(ctx, L, S, d) : chain_finalize_w<uint32_t>(ctx, L, S, d);
}

// ---- a group of samples in one launch chain (as dense_group_enqueue) --------------------------------------------------------
void bucket_lane_bytes(const psk_ctx *ctx, size_t max_len, size_t out[5])
{
    const size_t wb = bs_wide(ctx) ? 8 : 4, tile = bs_wide(ctx) ? (size_t)BsGeo<uint64_t>::TILE : (size_t)BsGeo<uint32_t>::TILE;
    out[0] = max_len * wb + 64;                                  // dc_part
    out[1] = (size_t)div_up(max_len, tile) * BS_NB * 4;          // dc_wgoff
    out[2] = (size_t)BS_SLOTS * (BS_NB + 16) * 4;                // dc_cnt
    out[3] = (size_t)(4 * BS_NB + 4) * 4;                        // dc_meta
    out[4] = (max_len + 8) * (wb + 4) + 64;                      // dc_mtemp
}

int bucket_group_enqueue(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, const uint64_t *clean_len, const uint64_t *n,
                         int count)
{
    (void)sample_idx;
    if (count < 1 || count > BS_GROUP) return psk_fail(ctx, PSK_EINVAL, "bad group size %d", count);
    return bs_wide(ctx) ? group_enqueue_w<uint64_t>(ctx, lanes, clean_len, n, count) : group_enqueue_w<uint32_t>(ctx, lanes, clean_len, n, count);
}

// after chain_finalize of every set of the group (which sized and allocated the lists): ONE packing launch
int bucket_group_compact(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, int count)
{
    return bs_wide(ctx) ? group_compact_w<uint64_t>(ctx, lanes, sample_idx, count) : group_compact_w<uint32_t>(ctx, lanes, sample_idx, count);
}

// the fall-back's input: the n_kept partitioned words of the sample on this set as u64 keys in `keys`
int bucket_fallback_keys(psk_ctx *ctx, CountLane &L, uint64_t n_kept, uint64_t *keys)
{
    if (n_kept) {
        if (bs_wide(ctx)) bs_expand_kernel<uint64_t><<<div_up(n_kept, 256), 256, 0, ctx->stream>>>(L.dc_part.as<uint64_t>(), n_kept, keys);
        else bs_expand_kernel<uint32_t><<<div_up(n_kept, 256), 256, 0, ctx->stream>>>(L.dc_part.as<uint32_t>(), n_kept, keys);
    }
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}
