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

namespace {

constexpr int DT_THREADS = 512;
constexpr int DT_SEG = 32;                       // window ends per thread
constexpr int DT_TILE = DT_THREADS * DT_SEG;     // 16,384 bases per workgroup
constexpr int DC_VALS = 1 << DC_VB;
constexpr uint32_t DC_SLOTS = 32;                // pre-zeroed counter slots per buffer set
constexpr size_t DC_DENSE_BYTES_MAX = 64ull << 30;
constexpr size_t DP_LDS_BYTES = (size_t)(DT_TILE + DC_MAX_NB + DC_MAX_NB / 2 + 16) * 4;   // 76 KB: two workgroups per CU
constexpr size_t DCNT_LDS_BYTES = (size_t)(DC_VALS / 2 + 16) * 4;                         // 64 KB + scan scratch

struct Roll32 {
    uint32_t fw, rc;
    int run;
};

__device__ __forceinline__ void roll32(Roll32 &r, uint32_t c, uint32_t mask, int rcshift, int k)
{
    if (c == '\n') {
        r.run = 0;
    } else {
        const uint32_t code = ((c >> 1) ^ (c >> 2)) & 3u;  // A/a 0, C/c 1, G/g 2, T/t/U/u 3
        r.fw = ((r.fw << 2) | code) & mask;
        r.rc = (r.rc >> 2) | ((3u - code) << rcshift);
        r.run = (r.run < k) ? r.run + 1 : k;
    }
}

// The thread's DT_SEG bytes at s and the 32 before them (what lies before the buffer counts as a break).
struct Seg {
    uint32_t cur[DT_SEG / 4], prev[8];
};

__device__ __forceinline__ void load_seg(Seg &g, const uint8_t *__restrict__ clean, uint64_t len, uint64_t s)
{
#pragma unroll
    for (int j = 0; j < DT_SEG / 4; j++) g.cur[j] = 0x0a0a0a0au;
#pragma unroll
    for (int j = 0; j < 8; j++) g.prev[j] = 0x0a0a0a0au;
    if (s < len) {
        const uint4 *p = reinterpret_cast<const uint4 *>(clean + s);
#pragma unroll
        for (int q = 0; q < DT_SEG / 16; q++) {
            const uint4 a = p[q];
            g.cur[4 * q] = a.x; g.cur[4 * q + 1] = a.y; g.cur[4 * q + 2] = a.z; g.cur[4 * q + 3] = a.w;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint64_t back = (uint64_t)(2 - q) * 16;
            if (s >= back) {
                const uint4 c = *reinterpret_cast<const uint4 *>(clean + s - back);
                g.prev[4 * q] = c.x; g.prev[4 * q + 1] = c.y; g.prev[4 * q + 2] = c.z; g.prev[4 * q + 3] = c.w;
            }
        }
    }
}

__device__ __forceinline__ Roll32 warm_up(const Seg &g, uint32_t mask, int rcshift, int k)
{
    Roll32 r{0, 0, 0};
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (j >= 32 - (k - 1)) roll32(r, (g.prev[j >> 2] >> ((j & 3) * 8)) & 0xffu, mask, rcshift, k);
    }
    return r;
}

// exclusive scan over a workgroup of DT_THREADS threads, one value per thread
__device__ __forceinline__ uint32_t block_scan512(uint32_t v, uint32_t *total, uint32_t *lds)
{
    return psk_block_excl_scan_u32<DT_THREADS>(v, total, lds);
}

__global__ __launch_bounds__(DT_THREADS) void dc_hist_kernel(const uint8_t *__restrict__ clean, uint64_t len, int k, uint32_t lo,
                                                              uint32_t hi, uint32_t b0, uint32_t nb, uint32_t *__restrict__ cnt,
                                                              uint32_t *__restrict__ wgoff)
{
    __shared__ uint32_t h[DC_MAX_NB];
    for (uint32_t d = threadIdx.x; d < nb; d += DT_THREADS) h[d] = 0;
    __syncthreads();
    const uint64_t s = ((uint64_t)blockIdx.x * DT_THREADS + threadIdx.x) * DT_SEG;
    const uint32_t mask = (1u << (2 * k)) - 1u;
    const int rcshift = 2 * (k - 1);
    Seg g;
    load_seg(g, clean, len, s);
    Roll32 r = warm_up(g, mask, rcshift, k);
#pragma unroll
    for (int j = 0; j < DT_SEG; j++) {
        roll32(r, (g.cur[j >> 2] >> ((j & 3) * 8)) & 0xffu, mask, rcshift, k);
        const uint32_t w = r.fw < r.rc ? r.fw : r.rc;
        if (s + j < len && r.run >= k && w >= lo && w < hi) atomicAdd(&h[(w >> DC_VB) - b0], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nb; d += DT_THREADS) {
        const uint32_t c = h[d];
        if (c) wgoff[(uint64_t)blockIdx.x * nb + d] = atomicAdd(&cnt[d], c);
    }
}

__global__ __launch_bounds__(DT_THREADS) void dc_partition_kernel(const uint8_t *__restrict__ clean, uint64_t len, int k, uint32_t lo,
                                                                   uint32_t hi, uint32_t b0, uint32_t nb,
                                                                   const uint32_t *__restrict__ cnt,
                                                                   const uint32_t *__restrict__ wgoff,
                                                                   uint32_t *__restrict__ base_out, uint16_t *__restrict__ part)
{
    extern __shared__ uint32_t dyn_lds[];                 // DP_LDS_BYTES
    uint32_t *stage = dyn_lds;                            // 64 KB: (bucket << 15 | value), bucketed
    uint32_t *h = stage + DT_TILE;                        // counts, then (start of this tile's range in the bucket) - (local start)
    uint16_t *lstart = reinterpret_cast<uint16_t *>(h + DC_MAX_NB);
    uint32_t *scan_lds = h + DC_MAX_NB + DC_MAX_NB / 2;
    for (uint32_t d = threadIdx.x; d < nb; d += DT_THREADS) h[d] = 0;
    __syncthreads();
    const uint64_t s = ((uint64_t)blockIdx.x * DT_THREADS + threadIdx.x) * DT_SEG;
    const uint32_t mask = (1u << (2 * k)) - 1u;
    const int rcshift = 2 * (k - 1);
    Seg g;
    load_seg(g, clean, len, s);
    Roll32 r = warm_up(g, mask, rcshift, k);
    uint32_t wv[DT_SEG];   // word relative to the first bucket, or ~0
    uint16_t rk[DT_SEG];   // rank inside (tile, bucket)
#pragma unroll
    for (int j = 0; j < DT_SEG; j++) {
        roll32(r, (g.cur[j >> 2] >> ((j & 3) * 8)) & 0xffu, mask, rcshift, k);
        const uint32_t w = r.fw < r.rc ? r.fw : r.rc;
        wv[j] = 0xffffffffu;
        rk[j] = 0;
        if (s + j < len && r.run >= k && w >= lo && w < hi) {
            wv[j] = w - (b0 << DC_VB);
            rk[j] = (uint16_t)atomicAdd(&h[wv[j] >> DC_VB], 1u);
        }
    }
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
                h[d] = gex + (c4[e] ? wgoff[(uint64_t)blockIdx.x * nb + d] : 0u) - lex;
                if (blockIdx.x == 0) base_out[d] = gex;
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

// One workgroup per bucket.  bitmap: this sample's, DC_BUCKET_WORDS u64 per bucket.  mt_w / mt_f: multi-count
// entries before compaction; bucket b writes from slot base[b] / 2 on (it has at most cnt[b] / 2 of them, and the
// buckets' key ranges are laid out back to back, so the slots of different buckets cannot meet).
__global__ __launch_bounds__(DT_THREADS) void dc_count_kernel(const uint16_t *__restrict__ part, const uint32_t *__restrict__ cnt,
                                                               const uint32_t *__restrict__ base, uint32_t b0,
                                                               uint64_t *__restrict__ bitmap, uint32_t *__restrict__ mt_w,
                                                               uint32_t *__restrict__ mt_f, uint32_t *__restrict__ uniq_out,
                                                               uint32_t *__restrict__ multi_out)
{
    extern __shared__ uint32_t dyn_lds[];   // DCNT_LDS_BYTES
    uint32_t *tbl = dyn_lds;                // 64 KB
    uint32_t *scan_lds = tbl + DC_VALS / 2;
    const uint32_t b = blockIdx.x, t = threadIdx.x;
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
        for (uint32_t i = t; i < n; i += DT_THREADS) {
            const uint32_t v = part[off + i];
            atomicAdd(&tbl[((v >> 1) & 31u) * DT_THREADS + (v >> 6)], 1u << ((v & 1u) * 16));
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

// totals[0] = keys kept, [1] = distinct words, [2] = multi-count entries; moff = exclusive scan of multi
__global__ __launch_bounds__(1024) void dc_totals_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ uniq,
                                                         const uint32_t *__restrict__ multi, uint32_t nb,
                                                         uint32_t *__restrict__ moff, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t lds[16];
    const uint32_t t = threadIdx.x;
    uint32_t c = 0, u = 0, m[2] = {0, 0};
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t d = t * 2 + e;
        if (d < nb) { c += cnt[d]; u += uniq[d]; m[e] = multi[d]; }
    }
    uint32_t ctot, utot, mtot;
    psk_block_excl_scan_u32<1024>(c, &ctot, lds);
    psk_block_excl_scan_u32<1024>(u, &utot, lds);
    uint32_t mex = psk_block_excl_scan_u32<1024>(m[0] + m[1], &mtot, lds);
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t d = t * 2 + e;
        if (d < nb) moff[d] = mex;
        mex += m[e];
    }
    if (t == 0) { totals[0] = ctot; totals[1] = utot; totals[2] = mtot; totals[3] = 0; }
}

__global__ void dc_compact_kernel(const uint32_t *__restrict__ mt_w, const uint32_t *__restrict__ mt_f,
                                  const uint32_t *__restrict__ base, const uint32_t *__restrict__ multi,
                                  const uint32_t *__restrict__ moff, uint32_t *__restrict__ dst_w, uint32_t *__restrict__ dst_f)
{
    const uint32_t b = blockIdx.x, n = multi[b], src = base[b] / 2, dst = moff[b];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        dst_w[dst + i] = mt_w[src + i];
        dst_f[dst + i] = mt_f[src + i];
    }
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

int dense_chain_enqueue(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n)
{
    SampleList &S = ctx->lists[sample_idx];
    const uint32_t nb = ctx->dense_nb, b0 = ctx->dense_b0;
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
    uint32_t *cnt = L.dc_cnt.as<uint32_t>() + (size_t)DC_MAX_NB * L.dc_slot++;
    uint32_t *meta = L.dc_meta.as<uint32_t>();
    uint32_t *base = meta, *uniq = meta + DC_MAX_NB, *multi = meta + 2 * DC_MAX_NB, *moff = meta + 3 * DC_MAX_NB,
             *totals = meta + 4 * DC_MAX_NB;
    uint32_t *mt_w = L.dc_mtemp.as<uint32_t>(), *mt_f = mt_w + (n / 2 + 2);
    const uint8_t *clean = L.raw.as<uint8_t>();
    dc_hist_kernel<<<n_tiles, DT_THREADS, 0, ctx->stream>>>(clean, clean_len, ctx->k, lo, hi, b0, nb, cnt, L.dc_wgoff.as<uint32_t>());
    PSK_HIP(ctx, hipGetLastError());
    static bool lds_set = false;
    if (!lds_set) {
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(dc_partition_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)DP_LDS_BYTES));
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(dc_count_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)DCNT_LDS_BYTES));
        lds_set = true;
    }
    dc_partition_kernel<<<n_tiles, DT_THREADS, DP_LDS_BYTES, ctx->stream>>>(clean, clean_len, ctx->k, lo, hi, b0, nb, cnt,
                                                                  L.dc_wgoff.as<uint32_t>(), base, L.dc_part.as<uint16_t>());
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipEventRecord(L.raw_free, ctx->stream));
    L.raw_used = true;
    dc_count_kernel<<<nb, DT_THREADS, DCNT_LDS_BYTES, ctx->stream>>>(L.dc_part.as<uint16_t>(), cnt, base, b0, S.bitmap, mt_w, mt_f, uniq, multi);
    PSK_HIP(ctx, hipGetLastError());
    dc_totals_kernel<<<1, 1024, 0, ctx->stream>>>(cnt, uniq, multi, nb, moff, totals);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(L.pinned_cnt, totals, 16, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipEventRecord(L.done, ctx->stream));
    L.dense = true;
    return PSK_OK;
}

// after hipEventSynchronize(L.done): sizes and fills the arena block of the multi-count entries
int dense_chain_finalize(psk_ctx *ctx, CountLane &L, uint64_t *n_kept, uint64_t *n_unique)
{
    SampleList &S = ctx->lists[L.sample];
    *n_kept = L.pinned_cnt[0];
    *n_unique = L.pinned_cnt[1];
    const uint64_t nm = L.pinned_cnt[2];
    S.n_multi = nm;
    if (nm) {
        PSK_TRY(arena_alloc(ctx, nm * 4, (void **)&S.mwords));
        PSK_TRY(arena_alloc(ctx, nm * 4, (void **)&S.mfreqs));
        uint32_t *meta = L.dc_meta.as<uint32_t>();
        uint32_t *mt_w = L.dc_mtemp.as<uint32_t>(), *mt_f = mt_w + (L.n / 2 + 2);
        dc_compact_kernel<<<ctx->dense_nb, 128, 0, ctx->stream>>>(mt_w, mt_f, meta, meta + 2 * DC_MAX_NB, meta + 3 * DC_MAX_NB,
                                                                 S.mwords, S.mfreqs);
        PSK_HIP(ctx, hipGetLastError());
    }
    L.dense = false;
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
