// a2 + a3 for k >= 14 (config 3: 2,048 samples, k = 16): union + presence matrix by a STREAMING MERGE of the
// per-sample lists, which are already sorted -- no pair sort (get_feature_vector / get_union / map_samples,
// modeling.py:317-380: the reference merges the lists on disk with glistcompare -u and maps every sample back with
// glistquery -l).
//
// r02's route packed every (word, sample) pair into a u64 and radix-sorted 1.28 G of them per slab: 111 ms and
// ~424 GB of HBM traffic for 21 GB of algorithmic bytes.  Here the slab's word range is cut into tiles at
// quantiles of a pilot of the lists (~8,192 pairs each; a tile spans at most PM_BMW * 64 word values), consecutive
// tiles form a range, and ONE workgroup streams a range for a group of 1,024 samples: lane l of wave v owns sample
// 1024 g + 64 v + l, finds its cursor once (one binary search per lane per workgroup) and from then on only moves
// forward -- the end of a tile is the start of the next, so there is no tile table at all.  Inside a tile a wave runs a
// 64-way merge: m = minimum of the lanes' current words (DPP min-reduce), ballot(current == m) IS the 64 presence bits
// of word m for the wave's samples, the lanes that hit advance.  Words shared by many samples (the ancestral k-mers:
// 95 % of the pairs of config 3) cost one iteration per wave instead of 64 atomics.
//   pass 1  pm_mark:   per tile an LDS bitmap of the word values that occur -> global occupancy bitmap (1 bit per
//                      word value of the slab: 35-190 MB at config 3)
//   (scan)  popcounts of the bitmap words -> exclusive scan = rank of every word value = its row; M = total
//   pass 2  pm_fill:   the same stream again; row of m = rank[(m - lo) >> 6] + popcount of the lower bits; wave v
//                      stores its ballot into column v of the tile's LDS block (every (row, column) is written at
//                      most once: plain stores, no atomics); rows are copied out coalesced, 128 B per row and group;
//                      group 0 also expands the bitmap into the union words
// Traffic: lists read twice (2 x 8 B per pair) + matrix written once + the bitmap; no (word, sample) pair is ever
// written.  Rows come out in ascending word order (glistcompare's).
#include "dev_utils.h"
#include "psk_internal.h"

#include <algorithm>
#include <chrono>

namespace {

#ifndef PSK_PM_GROUP
#define PSK_PM_GROUP 1024
#endif
constexpr int PM_GROUP = PSK_PM_GROUP;  // samples per workgroup: 16 waves x 64 lanes
constexpr int PM_COLS = PM_GROUP / 64;  // u64 columns of a row one group writes
constexpr uint32_t PM_BMW = 2048;       // most bitmap words (64 word values each) a tile may span: 16 KB + 8 KB of ranks
constexpr uint64_t PM_SENT = ~0ull;     // "no word" in the pilot: beyond any canonical word
constexpr size_t PM_LDS_MAX = 144 * 1024;

struct PmList {
    const uint64_t *words;
    uint64_t n;
};

typedef unsigned long long pm_u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) pm_u64x2 *pm_gvec;   // global address space: global_load_dwordx4, not flat

__device__ __forceinline__ uint32_t pm_dpp(uint32_t v, const int tag)
{
    int x = (int)v;
    switch (tag) {  // constant-folded (the builtin wants an immediate control word)
    case 0: x = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: x = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: x = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
    default: x = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); break;  // row_mirror
    }
    return (uint32_t)x;
}
__device__ __forceinline__ uint64_t pm_dpp(uint64_t v, const int tag)
{
    return ((uint64_t)pm_dpp((uint32_t)(v >> 32), tag) << 32) | pm_dpp((uint32_t)v, tag);
}
__device__ __forceinline__ uint32_t pm_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t pm_readlane(uint64_t v, int l) { return psk_readlane_u64(v, l); }

// minimum over the 64 lanes, wave-uniform; all lanes active.  Quad steps, then the mirrors act as xor-4 / xor-8
// butterflies on values that are already uniform per quad / per 8; the four rows meet through scalar lane reads.
// 32-bit words: the minimum with the DPP operand inside the v_min (one instruction per butterfly step instead of a
// copy, a DPP move and the min).  A DPP operand may not be read for two wait states after the VALU write that
// produced it (s_nop 1; the assembler does not add it inside inline asm).
__device__ __forceinline__ uint32_t pm_wave_min(uint32_t v)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    const uint32_t a = pm_readlane(v, 0), b = pm_readlane(v, 16), c = pm_readlane(v, 32), d = pm_readlane(v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

template <typename W>
__device__ __forceinline__ W pm_wave_min(W v)
{
    W o;
    o = pm_dpp(v, 0); v = o < v ? o : v;
    o = pm_dpp(v, 1); v = o < v ? o : v;
    o = pm_dpp(v, 2); v = o < v ? o : v;
    o = pm_dpp(v, 3); v = o < v ? o : v;
    const W a = pm_readlane(v, 0), b = pm_readlane(v, 16), c = pm_readlane(v, 32), d = pm_readlane(v, 48);
    const W ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// The minimum when lane 0 already stands on it -- every word all the wave's samples share, 95 % of the iterations at
// config 3 --: one lane read and one compare instead of the butterflies (4 DPP steps with their wait states, 4 lane reads)
template <typename W>
__device__ __forceinline__ W pm_wave_min_guess(W cand)
{
    const W m0 = pm_readlane(cand, 0);
    if (__ballot(cand < m0) == 0ull) return m0;
    return pm_wave_min(cand);
}

// A lane's cursor into its list.  W = uint32_t: words relative to the slab's base (word spaces of up to 2^32 values:
// every k <= 16), W = uint64_t otherwise; the all-ones value means "no word" (never a canonical word: the reverse
// complement of T...T is A...A).
// The lane holds a window of two aligned blocks of B words in registers (r[0 .. 2B)), the index i of its current word
// inside the window, and the block behind the window (pv), requested ahead.  Loads happen ONLY in refill(), which the
// whole wave runs when some lane has used its window up: every lane that is past its first block moves the window one
// block on and requests the block after, so after a refill every lane has at least B + 1 words ahead, the merge loop
// runs at least B + 1 iterations without touching memory, and a requested block has that long to arrive.  (First cut:
// each lane prefetched its next block on its own, in the loop; the compiler has to wait for ALL outstanding loads before
// the next use of any of them, so with the lanes out of phase nearly every iteration paid a full memory latency: 13.8 +
// 21.2 ms for config 3's slab.)  B = 8: a block is 64 bytes -- with 32-byte blocks every 128-byte line of a list went
// through the L2 four times (each lane comes back to its line four refills later, 8 MB of lines in flight per XCD
// against 4 MB of L2): 73 GB of L2 fills for the 20.5 GB the two passes read (PMC, r03).
#ifndef PSK_PM_BLOCK
#define PSK_PM_BLOCK 8
#endif
#ifndef PSK_PM_NT
#define PSK_PM_NT 0
#endif
// r04: for 32-bit words the window is a RING IN LDS (2 B words per lane, word-major: slot j of every lane of the workgroup
// side by side, so that lanes standing on different slots still hit 32 different banks).  The window in registers cost the
// merge loop a 15-deep tree of v_bfi per advance (a run-time index into registers does not exist) and B moves per
// refill; in LDS an advance is one ds_read.  The loops are VALU-issue bound (~75 instructions per wave iteration, 43 M
// iterations per pass at config 3: pm_mark 6.8 + pm_fill 14.2 ms, r03); 64-bit words (k = 17 slabs) keep the registers:
// their ring would be 128 KB.
template <typename W, int B = PSK_PM_BLOCK, bool RING_ = (sizeof(W) == 4)>
struct PmCursor {
    static constexpr W SENT = (W)~(W)0;
    static constexpr bool RING = RING_;   // (r05: 64-bit words ride the ring too where the kernel has the LDS for it -- the bitmap-free pass 1)
    static constexpr int NREG = RING ? 1 : 2 * B;
    W *ring;               // RING: this lane's slot 0; slot j at ring[j * stride]
    uint32_t stride, off;  // RING: lanes per workgroup; the slot of window position 0 (0 or B: the halves swap roles at a refill)
    const uint64_t *w;     // the list, shifted down so that block b = w[B b .. B b + B - 1] is aligned to its size
    uint32_t first, end;   // the list's words are w[first .. end)
    uint32_t last_blk;     // block of the list's last word
    uint32_t blk;          // block index of r[0 .. B)
    uint32_t i;            // current word = r[i]; i == 2 B: the window is used up
    uint64_t base;
    W r[NREG], cur;
    W nxt;                 // RING: the word behind the current one, read an iteration ahead (an advance never waits for LDS)
    pm_u64x2 pv[B / 2];    // the block behind the window as it was loaded, requested one refill ahead (see refill); it is
                           // converted only when it moves into the window -- touching it earlier would make the
                           // compiler wait for the load right where it was issued

    __device__ __forceinline__ W conv(uint64_t x, uint32_t idx) const
    {
        // the test of the high half is always true for W = uint32_t (the words of an eligible slab are within 2^32 of its
        // base); it keeps the high registers of a requested block alive until the block is taken -- dead, the compiler
        // reuses them as temporaries right behind the load and has to wait for the load first
        const uint64_t d = x - base;
        return (idx >= first && idx < end && (sizeof(W) == 8 || (d >> 32) == 0)) ? (W)d : SENT;
    }
    // raw block b -> pv (not waited for).  Always a load, never a branch: a block beyond the list is read from the
    // list's last block instead (take() turns indices beyond the end into "no word" whatever was read), because a
    // conditional assignment makes the compiler load into temporaries and copy -- i.e. wait -- on the spot.  An aligned
    // block that holds at least one word of the list lies inside the list's (256-byte granular) allocation.
    __device__ __forceinline__ void request(uint32_t b)
    {
        const uint32_t bb = b < last_blk ? b : last_blk;
        const pm_gvec p = (pm_gvec)(uintptr_t)(w + B * (size_t)bb);
#pragma unroll
        for (int k = 0; k < B / 2; k++) pv[k] = PSK_PM_NT ? __builtin_nontemporal_load(&p[k]) : p[k];
    }
    // block b, requested before, into window positions [B half, B half + B)
    __device__ __forceinline__ void take(uint32_t b, int half)
    {
#pragma unroll
        for (int k = 0; k < B / 2; k++) {
            const W x0 = conv(pv[k].x, B * b + 2 * k), x1 = conv(pv[k].y, B * b + 2 * k + 1);   // beyond the end: SENT
            if constexpr (RING) {
                const uint32_t s0 = (uint32_t)(B * half + 2 * k + off) & (2 * B - 1);
                ring[s0 * stride] = x0;
                ring[(s0 + 1) * stride] = x1;     // (2 k + off is even: no wrap inside the pair)
            } else {
                r[(B * half + 2 * k) % NREG] = x0;
                r[(B * half + 2 * k + 1) % NREG] = x1;
            }
        }
    }
    // v[i] for i < N by a binary tree of selects: bit 0 of i halves the candidates, then bit 1, ... (N - 1 v_cndmask; written
    // as a recursion over constant sizes -- as a loop over a run-time step the compiler fell back to comparing i with every
    // index for every element: 700 instructions per advance)
    template <int N>
    static __device__ __forceinline__ W pick(const W *v, uint32_t i)
    {
        if constexpr (N == 1) {
            return v[0];
        } else {
            // (b & m) | (a & ~m) with m = all ones when the bit is set: one v_bfi_b32.  Written as `bit ? b : a` the compiler
            // turns the pair into an indexed read of a stack array -- scratch memory inside the merge loop
            W h[N / 2];
            const W m = (W)0 - (W)(i & 1u);
#pragma unroll
            for (int k = 0; k < N / 2; k++) h[k] = (v[2 * k + 1] & m) | (v[2 * k] & ~m);
            return pick<N / 2>(h, i >> 1);
        }
    }
    // i == 2 B reads position 0: the loop refills before it looks
    __device__ __forceinline__ void select()
    {
        if constexpr (RING) {
            cur = ring[((i + off) & (2 * B - 1)) * stride];
            nxt = ring[((i + 1 + off) & (2 * B - 1)) * stride];   // (i + 1 == 2 B: some slot; it only ever becomes `cur` of a dry window)
        } else cur = pick<NREG>(r, i);
    }
    // pos: index into the list of the word to stand on (pos == n: at the end)
    // words == nullptr or n == 0 (no sample in this lane, an empty list): `spare` (64-byte aligned, 64 bytes) is read instead
    __device__ __forceinline__ void seek(const uint64_t *words, uint32_t n, uint32_t pos, uint64_t base_, const uint64_t *spare)
    {
        if (!words || n == 0) { words = spare; n = 0; }
        const uint32_t mis = (uint32_t)(((uintptr_t)words >> 3) & (B - 1));
        w = words - mis; first = mis; end = n + mis; base = base_;
        last_blk = end ? (end - 1) / B : 0;
        const uint32_t q = pos + mis;
        blk = q / B; i = q % B; off = 0;
        request(blk); take(blk, 0);
        request(blk + 1); take(blk + 1, 1);
        request(blk + 2);
        select();
    }
    // RING: where this lane's ring lives (before the first seek)
    __device__ __forceinline__ void attach(W *lane_slot0, uint32_t lanes) { ring = lane_slot0; stride = lanes; off = 0; }
    __device__ __forceinline__ uint32_t position() const { return B * blk + i - first; }   // index into the list of the current word
    __device__ __forceinline__ bool dry() const { return i >= 2 * B; }
    // wave-uniform call.  The block that moves into the window was requested at the lane's PREVIOUS refill (or seek), at
    // least B + 1 iterations of the merge loop ago; the one requested here is not waited for until the next refill.
    __device__ __forceinline__ void refill()
    {
        if (i >= B) {
            if constexpr (RING) off ^= B;   // the second half becomes the first; the block behind it goes where the first was
            else {
#pragma unroll
                for (int k = 0; k < B; k++) r[k % NREG] = r[(B + k) % NREG];
            }
            blk++; i -= B;
            take(blk + 1, 1);
            request(blk + 2);
            select();
        }
    }
    __device__ __forceinline__ void advance()
    {
        i++;
        if constexpr (RING) {
            cur = nxt;
            nxt = ring[((i + 1 + off) & (2 * B - 1)) * stride];
        } else select();   // i == 2 B: some value of the window; the loop refills before it looks at `cur` again
    }
};

__device__ __forceinline__ uint32_t pm_lower_bound(const uint64_t *w, uint32_t n, uint64_t key)
{
    uint32_t a = 0, b = n;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (w[mid] < key) a = mid + 1; else b = mid;
    }
    return a;
}

// pass 1: the word values that occur, per tile in LDS, then into the global occupancy bitmap
// r04: the pass also leaves what its waves found -- (word, ballot) of every iteration, 64 at a time, lane j the j-th -- as
// RECORDS in chunks of a pool (PmRec): pass 2 then does not merge the lists again, it replays the records (pm_replay_kernel:
// a wave per chunk, lane j looks up the row of its word and stores its ballot).  A chunk is claimed with one returning
// atomic, asked for a chunk ahead (its latency is 64 iterations old when the id is needed).  The pool holds a twelfth of
// the pairs; a data set whose samples share too little for that (iterations ~ pairs) overflows it, the flag is set, the
// bitmap is complete all the same and the build takes the merging pass 2 (pm_fill_kernel) as before.
constexpr int PM_REC_REGIONS = 16;   // the pool is cut into regions with a counter each (ONE counter serialises: ~11 ns per
constexpr int PM_REC_BLOCK = 8;      // returning atomic, 1 M claims = 11 ms -- pm_mark 4.7 -> 13 ms); a claim is 8 chunks
constexpr int PM_REC_CTR_STRIDE = 32;   // u32 between two counters: a 128-byte line each
struct PmRec {
    uint32_t *ctr;          // [PM_REC_CTR_STRIDE r] chunks claimed in region r; [PM_REC_CTR_STRIDE PM_REC_REGIONS] overflow
    uint2 *hdr;             // per chunk {column (wave of 64 samples), records}; zeroed before the launch
    void *words;            // W[64] per chunk
    unsigned long long *masks;   // u64[64] per chunk
    uint32_t region_chunks; // chunks per region; 0: no records
};

// BITMAP = false (r05, word spaces beyond 2^34 values: k >= 18): no occupancy bitmap -- a bit per word VALUE is what such a
// space cannot have -- the pass only leaves its records; the union is then the sorted distinct record words and a record's
// row its rank among them (build_presence_merge_wide below).
template <typename W, bool BITMAP = true>
__global__ __launch_bounds__(PM_GROUP) void pm_mark_kernel(const PmList *__restrict__ lists, int n_samples,
                                                           const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                                           uint32_t tiles_per_range, uint64_t base,
                                                           unsigned long long *__restrict__ gbm, int single_group,
                                                           const uint64_t *__restrict__ spare, const PmRec rec)
{
    __shared__ unsigned long long bm[BITMAP ? PM_BMW : 1];
    extern __shared__ __attribute__((aligned(16))) unsigned long long pm_ring_lds[];   // the lanes' windows (32-bit words; 64-bit ones without the bitmap: 128 KB)
    typedef PmCursor<W, PSK_PM_BLOCK, (sizeof(W) == 4 || !BITMAP)> Cur;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.y * (BITMAP ? PM_GROUP : (int)blockDim.x) + threadIdx.x;   // (bitmap-free: the group is the workgroup, 512 lanes with 64-bit cursors)
    const uint32_t t0 = blockIdx.x * tiles_per_range;
    const uint32_t t1 = t0 + tiles_per_range < n_tiles ? t0 + tiles_per_range : n_tiles;
    // bitmap-free pass with 32-bit cursors (r05): the words of THIS range relative to its first bound -- the host takes this
    // instantiation when every range of the launch spans less than 2^32 - 1 word values (k = 21 under a slab filter: 2^29); the
    // records carry the absolute 64-bit words again
    const uint64_t base_r = (!BITMAP && sizeof(W) == 4) ? bounds[t0] : base;
    Cur cur;
    if constexpr (Cur::RING) cur.attach(reinterpret_cast<W *>(pm_ring_lds) + threadIdx.x, blockDim.x);
    if (s < n_samples) {
        const PmList L = lists[s];
        cur.seek(L.words, (uint32_t)L.n, pm_lower_bound(L.words, (uint32_t)L.n, bounds[t0]), base_r, spare);
    } else {
        cur.seek(nullptr, 0, 0, base_r, spare);
    }
    // records: lane j keeps the j-th (word, ballot) since the last chunk went out; they run on across the tiles
    const bool recs = rec.region_chunks != 0;
    const uint32_t col = (uint32_t)(s >> 6);
    W rec_m = 0;
    uint64_t rec_mask = 0;
    int rcnt = 0;
    // the chunk being filled = ch_cur (region-relative; a claim is PM_REC_BLOCK chunks); lane 0 holds the next claim, asked for
    // when this one was begun (not waited for until it is needed)
    const uint32_t region = (blockIdx.x + blockIdx.y) % PM_REC_REGIONS;
    uint32_t *my_ctr = rec.ctr + region * PM_REC_CTR_STRIDE;
    const size_t region0 = (size_t)region * rec.region_chunks;
    uint32_t ch_cur = 0, ch_ahead = 0;
    if (recs) {
        if (lane == 0) ch_cur = atomicAdd(my_ctr, (uint32_t)PM_REC_BLOCK);
        ch_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch_cur);
        if (lane == 0) ch_ahead = atomicAdd(my_ctr, (uint32_t)PM_REC_BLOCK);
    }
    auto rec_flush = [&](int have) __attribute__((always_inline)) {
        if (ch_cur < rec.region_chunks) {
            const size_t ch = region0 + ch_cur;
            if (lane < have) {
                if (BITMAP) reinterpret_cast<W *>(rec.words)[ch * 64 + lane] = rec_m;
                else reinterpret_cast<uint64_t *>(rec.words)[ch * 64 + lane] = base_r + (uint64_t)rec_m;
                rec.masks[ch * 64 + lane] = rec_mask;
            }
            if (lane == 0) rec.hdr[ch] = make_uint2(col, (uint32_t)have);
        } else if (lane == 0 && have) rec.ctr[PM_REC_REGIONS * PM_REC_CTR_STRIDE] = 1u;   // the region is full: pass 2 merges again
    };
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        const W lo_w = (W)(lo - base_r), hi_w = (W)(hi - base_r - 1);   // inclusive upper end: hi - base may be 2^32
        const uint32_t nbw = BITMAP ? (uint32_t)((hi - lo + 63) >> 6) : 0u;
        if (BITMAP) {
            for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) bm[i] = 0;
            __syncthreads();
        }
        // the words found go into the bitmap 64 at a time, lane j the j-th of them (one atomic of one lane per iteration was a
        // wave instruction per word; this is one per 64)
        uint32_t my_v = 0;
        int cnt = 0;
        for (;;) {
            if (__any(cur.dry())) cur.refill();
            const W cand = cur.cur <= hi_w ? cur.cur : Cur::SENT;
            const W m = pm_wave_min_guess(cand);
            if (m == Cur::SENT) break;
            const bool hit = cand == m;
            if (BITMAP) {
                const uint32_t v = (uint32_t)(m - lo_w);
                if (lane == cnt) my_v = v;
                if (++cnt == 64) {
                    atomicOr(&bm[my_v >> 6], 1ull << (my_v & 63));
                    cnt = 0;
                }
            }
            if (recs) {
                const uint64_t mask = __ballot(hit);
                if (lane == rcnt) { rec_m = m; rec_mask = mask; }
                if (++rcnt == 64) {
                    rec_flush(64);
                    rcnt = 0;
                    if ((++ch_cur & (PM_REC_BLOCK - 1)) == 0) {   // (claims are multiples of PM_REC_BLOCK)
                        ch_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch_ahead);
                        if (lane == 0) ch_ahead = atomicAdd(my_ctr, (uint32_t)PM_REC_BLOCK);
                    }
                }
            }
            if (hit) cur.advance();
        }
        if (BITMAP) {
            if (lane < cnt) atomicOr(&bm[my_v >> 6], 1ull << (my_v & 63));
            __syncthreads();
            unsigned long long *g = gbm + ((lo - base) >> 6);
            for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) {
                const unsigned long long v = bm[i];
                if (single_group) g[i] = v;
                else if (v) atomicOr(&g[i], v);
            }
            __syncthreads();
        }
    }
    if (recs) rec_flush(rcnt);   // (the chunk claimed ahead stays empty: its header is zero)
}

// ---- word spaces without a value bitmap (k >= 18): union and rows from the records ------------------------------------------------
// records of every chunk -> one dense key array (chunk c's records at off[c] ..): a wave per chunk
// (only the chunks a region has CLAIMED are looked at -- n_rel of them per region, the most any region claimed: the pool is three
// times that; dense index d = region * n_rel + rel stands for chunk region * region_chunks + rel)
__global__ __launch_bounds__(256) void pmw_chunk_counts_kernel(const uint2 *__restrict__ hdr, uint32_t region_chunks, uint32_t n_rel,
                                                               uint32_t *__restrict__ cnt)
{
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n = (uint64_t)n_rel * PM_REC_REGIONS;
    if (d < n) cnt[d] = hdr[(d / n_rel) * region_chunks + d % n_rel].y;
    else if (d == n) cnt[d] = 0;   // the scan's extra element: off[n] = records in all
}
// (keys relative to the slab's first word: the sort then runs over the bits of the slab's span, not of the whole space)
__global__ __launch_bounds__(256) void pmw_gather_kernel(const uint2 *__restrict__ hdr, const uint64_t *__restrict__ words, uint32_t region_chunks,
                                                         uint32_t n_rel, const uint32_t *__restrict__ off, uint64_t lo, uint64_t *__restrict__ keys)
{
    const uint32_t rel = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (rel >= n_rel) return;
    const uint64_t c = (uint64_t)blockIdx.y * region_chunks + rel, d = (uint64_t)blockIdx.y * n_rel + rel;
    if (lane < hdr[c].y) keys[(uint64_t)off[d] + lane] = words[c * 64 + lane] - lo;
}
// heads of the runs of equal keys (the distinct words); then, the heads ranked by a scan, the union
__global__ __launch_bounds__(256) void pmw_heads_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ head)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i - 1] != keys[i]) ? 1u : 0u;
    else if (i == n) head[i] = 0;
}
__global__ __launch_bounds__(256) void pmw_union_kernel(const uint64_t *__restrict__ keys, uint64_t n, const uint32_t *__restrict__ rank,
                                                        uint64_t lo, uint64_t *__restrict__ union_words)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0 || keys[i - 1] != keys[i])) union_words[rank[i]] = keys[i] + lo;
}
// cells[c] = first row whose word is >= lo + (c << shift): a record's row is then a search over one cell of the union
__global__ __launch_bounds__(256) void pmw_cells_kernel(const uint64_t *__restrict__ uw, uint32_t m, uint64_t lo, uint32_t shift, uint32_t n_cells,
                                                        uint32_t *__restrict__ cells)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cells) return;
    const uint64_t off = (uint64_t)c << shift, key = lo + off;
    uint32_t a = 0, b = m;
    if (c == n_cells || (shift && (off >> shift) != c) || key < off) a = m;   // beyond the last word value
    else
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (uw[mid] < key) a = mid + 1; else b = mid;
        }
    cells[c] = a;
}
// pass 2 from the records (as pm_replay_kernel): the row of a word = its place in the union, found inside its cell
__global__ __launch_bounds__(256) void pmw_replay_kernel(const PmRec rec, uint32_t n_rel, const uint64_t *__restrict__ uw,
                                                         const uint32_t *__restrict__ cells, uint64_t lo, uint32_t shift, uint32_t n_cells,
                                                         int wpr, uint64_t *__restrict__ bits)
{
    const uint32_t rel = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rel >= n_rel) return;
    const uint64_t ch = (uint64_t)blockIdx.y * rec.region_chunks + rel;
    const uint2 h = rec.hdr[ch];
    if ((uint32_t)lane >= h.y) return;
    const uint64_t m = reinterpret_cast<const uint64_t *>(rec.words)[ch * 64 + lane];
    const unsigned long long mask = rec.masks[ch * 64 + lane];
    uint64_t c = (m - lo) >> shift;
    if (c >= n_cells) c = n_cells - 1;
    uint32_t a = cells[c], b = cells[c + 1];
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (uw[mid] < m) a = mid + 1; else b = mid;
    }
    bits[(uint64_t)a * (uint64_t)wpr + h.x] = mask;
}

// pass 2 from the records: a wave per chunk; lane j: row of its word = rank of its bitmap word + the set bits below it, its
// ballot goes to its wave's column of that row.  The matrix has been zeroed: a (row, column) no record names stays 0.
template <typename W>
__global__ __launch_bounds__(256) void pm_replay_kernel(const PmRec rec, uint32_t n_chunks, const unsigned long long *__restrict__ gbm,
                                                        const uint32_t *__restrict__ rank, int wpr, uint64_t *__restrict__ bits)
{
    // n_chunks: the most any region has claimed; blockIdx.y: the region
    const uint32_t rel = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rel >= n_chunks) return;
    const size_t ch = (size_t)blockIdx.y * rec.region_chunks + rel;
    const uint2 h = rec.hdr[ch];
    if ((uint32_t)lane >= h.y) return;
    const W m = reinterpret_cast<const W *>(rec.words)[ch * 64 + lane];
    const unsigned long long mask = rec.masks[ch * 64 + lane];
    const uint64_t wq = (uint64_t)m >> 6;
    const uint64_t r = (uint64_t)rank[wq] + (uint32_t)__popcll(gbm[wq] & ((1ull << ((uint32_t)m & 63)) - 1ull));
    bits[r * (uint64_t)wpr + h.x] = mask;
}

// the union words: the set bits of the occupancy bitmap, in order
__global__ void pm_union_kernel(const unsigned long long *__restrict__ gbm, const uint32_t *__restrict__ rank, uint64_t n_words,
                                uint64_t base, uint64_t *__restrict__ union_words)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    unsigned long long v = gbm[i];
    uint32_t r = rank[i];
    while (v) {
        union_words[r++] = base + (i << 6) + (uint64_t)__builtin_ctzll(v);
        v &= v - 1;
    }
}

__global__ void pm_popcount_kernel(const unsigned long long *__restrict__ gbm, uint64_t n_words, uint32_t *__restrict__ cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) cnt[i] = (uint32_t)__popcll(gbm[i]);
    else if (i == n_words) cnt[i] = 0;   // the scan's extra element: rank[n_words] = M
}

// rows of every tile, from the ranks at its two ends
__global__ void pm_tile_rows_kernel(const uint32_t *__restrict__ rank, const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                    uint64_t base, uint32_t *__restrict__ rows)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t a = (bounds[t] - base) >> 6, b = (bounds[t + 1] - base + 63) >> 6;
    rows[t] = rank[b] - rank[a];
}

#ifdef PSK_PM_STATS
__device__ unsigned long long pm_stats[16];
#define PM_ST(k) { if (lane == 0) { const unsigned long long now = wall_clock64(); st[k] += now - st_c; st_c = now; } }
#else
#define PM_ST(k)
#endif
// pass 2: the same stream; every wave stores its ballots into its column of the tile's block
#ifndef PSK_PM_FILL_WGS
#define PSK_PM_FILL_WGS 1
#endif
template <typename W>
__global__ __launch_bounds__(PM_GROUP, PSK_PM_FILL_WGS) void pm_fill_kernel(const PmList *__restrict__ lists, int n_samples, int wpr,
                                                           const uint64_t *__restrict__ bounds, uint32_t n_tiles,
                                                           uint32_t tiles_per_range, uint64_t base,
                                                           const unsigned long long *__restrict__ gbm,
                                                           const uint32_t *__restrict__ rank, uint32_t r_cap,
                                                           uint32_t bmw_max, uint64_t *__restrict__ union_words,
                                                           uint64_t *__restrict__ bits, const uint64_t *__restrict__ spare)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long pm_lds_all[];
    // (32-bit words) the lanes' windows first: 2 B words per lane
    unsigned long long *pm_lds = pm_lds_all + (PmCursor<W>::RING ? (size_t)blockDim.x * PSK_PM_BLOCK * 2 * sizeof(W) / 8 : 0);
    unsigned long long *bm = pm_lds;                                        // bmw_max
    uint32_t *rk = reinterpret_cast<uint32_t *>(pm_lds + bmw_max);          // bmw_max + 1 (padded to even)
    unsigned long long *blk = pm_lds + bmw_max + ((bmw_max + 2) >> 1);      // rows x cols of the batch, row-major (r04: the copy-out reads it with consecutive threads on consecutive words; column-major, rows a multiple of 16 put a wave's 16 columns on one bank)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = blockIdx.y;
    const int s = group * PM_GROUP + threadIdx.x;
    const int cols = wpr - group * PM_COLS < PM_COLS ? wpr - group * PM_COLS : PM_COLS;   // u64 words of a row this group writes
    const uint32_t t0 = blockIdx.x * tiles_per_range;
    const uint32_t t1 = t0 + tiles_per_range < n_tiles ? t0 + tiles_per_range : n_tiles;
    const uint64_t *lw = nullptr;
    uint32_t ln = 0;
    if (s < n_samples) { const PmList L = lists[s]; lw = L.words; ln = (uint32_t)L.n; }
    PmCursor<W> cur;
    if constexpr (PmCursor<W>::RING) cur.attach(reinterpret_cast<W *>(pm_lds_all) + threadIdx.x, blockDim.x);
    cur.seek(lw, ln, lw ? pm_lower_bound(lw, ln, bounds[t0]) : 0, base, spare);
#ifdef PSK_PM_STATS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_c = wall_clock64(), st_iter = 0;
#endif
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        const W lo_w = (W)(lo - base), hi_w = (W)(hi - base - 1);
        const uint32_t nbw = (uint32_t)((hi - lo + 63) >> 6);
        const uint64_t w0 = (lo - base) >> 6;
        for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) bm[i] = gbm[w0 + i];
        for (uint32_t i = threadIdx.x; i <= nbw; i += blockDim.x) rk[i] = rank[w0 + i];
        __syncthreads();
        PM_ST(0)
        const uint32_t row0 = rk[0], rows = rk[nbw] - row0;
        const uint32_t pos = rows > r_cap ? cur.position() : 0;   // where this lane's words of the tile begin (several batches only)
        for (uint32_t b0 = 0; b0 < rows; b0 += r_cap) {
            const uint32_t rb = rows - b0 < r_cap ? rows - b0 : r_cap;
            for (uint32_t i = threadIdx.x; i < rb * (uint32_t)cols; i += blockDim.x) blk[i] = 0;
            __syncthreads();
            PM_ST(1)
            if (b0 > 0) cur.seek(lw, ln, pos, base, spare);   // a tile with more rows than the block holds is streamed once per batch
            // A word's row = rank of its bitmap word + the set bits below it: two LDS reads, a popcount and the store -- per
            // iteration they were ~20 wave-uniform instructions behind an LDS round trip (pm_fill: 5.5 G instructions against
            // pm_mark's 3.6 G for the same stream, PMC r04).  Lane j keeps the j-th word and its ballot instead, and every 64
            // words the lanes look their rows up side by side.
            uint32_t my_i = 0;
            uint64_t my_mask = 0;
            int cnt = 0;
            auto flush = [&](int have) __attribute__((always_inline)) {
                if (lane < have && wave < cols) {
                    const uint32_t wq = my_i >> 6;
                    const uint32_t r = rk[wq] - row0 + (uint32_t)__popcll(bm[wq] & ((1ull << (my_i & 63)) - 1ull)) - b0;
                    if (r < rb) blk[r * (uint32_t)cols + (uint32_t)wave] = my_mask;   // r is unsigned: rows of earlier batches wrap
                }
            };
            for (;;) {
                if (__any(cur.dry())) cur.refill();
                const W cand = cur.cur <= hi_w ? cur.cur : PmCursor<W>::SENT;
                const W m = pm_wave_min_guess(cand);
                if (m == PmCursor<W>::SENT) break;
                const bool hit = cand == m;
                const uint64_t mask = __ballot(hit);
                if (lane == cnt) { my_i = (uint32_t)(m - lo_w); my_mask = mask; }
                if (++cnt == 64) { flush(64); cnt = 0; }
                if (hit) cur.advance();
#ifdef PSK_PM_STATS
                st_iter++;
#endif
            }
            flush(cnt);
            PM_ST(2)
            __syncthreads();
            PM_ST(3)
            uint64_t *dst = bits + (uint64_t)(row0 + b0) * wpr + (uint64_t)group * PM_COLS;
            if (cols == PM_COLS) {   // a full group: shifts instead of a division by a run-time value per element
                for (uint32_t e = threadIdx.x; e < rb * (uint32_t)PM_COLS; e += blockDim.x) {
                    const uint32_t r = e / PM_COLS, c = e % PM_COLS;
                    dst[(uint64_t)r * wpr + c] = blk[e];
                }
            } else {
                for (uint32_t e = threadIdx.x; e < rb * (uint32_t)cols; e += blockDim.x) {
                    const uint32_t r = e / (uint32_t)cols, c = e % (uint32_t)cols;
                    dst[(uint64_t)r * wpr + c] = blk[e];
                }
            }
            __syncthreads();
            PM_ST(4)
        }
        if (group == 0) {   // the union words of the tile: the set bits of its bitmap, in order
            for (uint32_t i = threadIdx.x; i < nbw; i += blockDim.x) {
                unsigned long long v = bm[i];
                uint32_t r = rk[i];
                while (v) {
                    union_words[r++] = lo + ((uint64_t)i << 6) + (uint64_t)__builtin_ctzll(v);
                    v &= v - 1;
                }
            }
        }
        __syncthreads();   // bm / rk are rewritten by the next tile
        PM_ST(5)
    }
#ifdef PSK_PM_STATS
    if (lane == 0) {
        for (int k = 0; k < 6; k++) atomicAdd(&pm_stats[k], st[k]);
        atomicAdd(&pm_stats[6], st_iter);
        atomicAdd(&pm_stats[7], 1ull);
    }
#endif
}

// evenly spaced entries of a few lists: the pilot the tile bounds are cut from
__global__ void pm_pilot_kernel(const PmList *__restrict__ lists, const int32_t *__restrict__ pick, int n_pick, uint32_t per_list,
                                uint64_t *__restrict__ out)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pick * per_list) return;
    const int li = (int)(g / per_list);
    const uint32_t j = (uint32_t)(g % per_list);
    const PmList L = lists[pick[li]];
    uint64_t v = PM_SENT;
    if (L.n) {
        uint64_t idx = (uint64_t)(((double)j + 0.5) * ((double)L.n / (double)per_list));
        if (idx >= L.n) idx = L.n - 1;
        v = L.words[idx];
    }
    out[g] = v;
}

}  // namespace

// Returns PSK_OK and sets *done = 1 when the merge build ran; *done = 0: not eligible, the caller takes the sort route.
int build_presence_merge(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done)
{
    *done = 0;
    if (getenv("PSK_NO_MERGE_PRESENCE")) return PSK_OK;
    const bool trace = getenv("PSK_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {   // PSK_TRACE: host-side phase times (each mark waits for the stream)
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[psk]   merge %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const int k = ctx->k, n = ctx->n_samples, wpr = ctx->wpr;
    if (2 * k > 40) return PSK_OK;
    const uint64_t space = 1ull << (2 * k);
    const uint64_t lo = ctx->slab_lo, hi = ctx->slab_hi ? ctx->slab_hi : space;
    const uint64_t base = lo & ~63ull, top = (hi + 63) & ~63ull;
    const uint64_t span = top - base;
    // one bit (+ half a rank byte) per word value of the slab: up to 2^34 values = 2 GB + 1 GB; rows are ranked in u32
    if (span > (1ull << 34)) return PSK_OK;
    // (r05) beyond k = 17 a narrow slab still has thousands of word values per word that occurs: the bitmap-free form is for those
    if (k >= 18 && span / 1024 > total_pairs) return PSK_OK;
    if (total_pairs >= (1ull << 32) && span / 2 + (1ull << k) >= (1ull << 32)) return PSK_OK;
    for (int i = 0; i < n; i++)
        if (ctx->lists[i].n_unique >= (1ull << 32)) return PSK_OK;
    // ---- tile bounds: pair quantiles of a pilot, no tile wider than PM_BMW bitmap words ------------------------------
    uint64_t pairs_per_tile = 8192;
    if (const char *e = getenv("PSK_MERGE_TILE_PAIRS")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 64) pairs_per_tile = v; }
    uint64_t want = (total_pairs + pairs_per_tile - 1) / pairs_per_tile;
    if (want < 1) want = 1;
    if (want > (1ull << 24)) want = 1ull << 24;
    const int n_pick = n < 64 ? n : 64;
    std::vector<int32_t> pick(n_pick);
    for (int i = 0; i < n_pick; i++) pick[i] = (int32_t)(((int64_t)i * n) / n_pick);
    uint64_t per_list = (4 * want + n_pick - 1) / n_pick;
    if (per_list < 64) per_list = 64;
    if (per_list > (1u << 22)) per_list = 1u << 22;
    const uint64_t n_pilot = per_list * (uint64_t)n_pick;
    std::vector<PmList> refs(n);
    for (int i = 0; i < n; i++) { refs[i].words = ctx->lists[i].words; refs[i].n = ctx->lists[i].n_unique; }
    PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)n * sizeof(PmList) + (size_t)n_pick * 4 + 64));
    PmList *d_refs = ctx->starts.as<PmList>();
    int32_t *d_pick = reinterpret_cast<int32_t *>(d_refs + n);
    PSK_HIP(ctx, hipMemcpyAsync(d_refs, refs.data(), (size_t)n * sizeof(PmList), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d_pick, pick.data(), (size_t)n_pick * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->keysA, n_pilot * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, n_pilot * 8));
    pm_pilot_kernel<<<div_up(n_pilot, 256), 256, 0, ctx->stream>>>(d_refs, d_pick, n_pick, (uint32_t)per_list, ctx->keysA.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    uint64_t *sorted = nullptr;
    PSK_TRY(dev_radix_sort_u64(ctx, ctx->keysA.as<uint64_t>(), ctx->keysB.as<uint64_t>(), n_pilot, 0, 64, &sorted));   // all 64 bits: empty lists contribute sentinels, which must sort last
    std::vector<uint64_t> pilot(n_pilot);
    PSK_HIP(ctx, hipMemcpyAsync(pilot.data(), sorted, n_pilot * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // also: refs / pick (host) have been copied
    mark("pilot + sort");
    while (!pilot.empty() && pilot.back() == PM_SENT) pilot.pop_back();   // empty lists contributed sentinels
    std::vector<uint64_t> bounds;
    bounds.push_back(base);
    const uint64_t max_gap = (uint64_t)PM_BMW * 64;
    auto push_bound = [&](uint64_t b) {   // b: multiple of 64 in (bounds.back(), top]; gaps wider than a tile may be are cut evenly
        uint64_t prev = bounds.back();
        if (b <= prev) return;
        const uint64_t gap = b - prev;
        if (gap > max_gap) {
            const uint64_t parts = (gap + max_gap - 1) / max_gap;
            for (uint64_t q = 1; q < parts; q++) {
                const uint64_t mid = (prev + (gap * q) / parts) & ~63ull;
                if (mid > bounds.back() && mid < b) bounds.push_back(mid);
            }
        }
        bounds.push_back(b);
    };
    if (!pilot.empty())
        for (uint64_t j = 1; j < want; j++) {
            const uint64_t v = pilot[(size_t)((pilot.size() * j) / want)] & ~63ull;
            if (v > base && v < top) push_bound(v);
        }
    push_bound(top);
    const uint64_t n_tiles64 = bounds.size() - 1;
    if (n_tiles64 > (1ull << 26)) return PSK_OK;   // a sparse word space cut into tiles of 2^17 values: not this route's case
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    uint32_t bmw_max = 1;
    for (uint32_t t = 0; t < n_tiles; t++) bmw_max = std::max<uint32_t>(bmw_max, (uint32_t)((bounds[t + 1] - bounds[t] + 63) >> 6));
    // ---- launch shape ---------------------------------------------------------------------------------------------
    const int n_groups = (n + PM_GROUP - 1) / PM_GROUP;
    const int threads = n >= PM_GROUP ? PM_GROUP : ((n + 63) / 64) * 64;
    uint64_t n_ranges = ((uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8 * (PM_GROUP / threads)) / n_groups;
    if (const char *e = getenv("PSK_MERGE_RANGES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1) n_ranges = v; }
    if (n_ranges > n_tiles) n_ranges = n_tiles;
    if (n_ranges < 1) n_ranges = 1;
    const uint32_t tiles_per_range = (uint32_t)((n_tiles + n_ranges - 1) / n_ranges);
    n_ranges = (n_tiles + tiles_per_range - 1) / tiles_per_range;
    // ---- buffers: bounds | occupancy bitmap | ranks | rows per tile ---------------------------------------------------
    const uint64_t n_bmw = span >> 6;
    PSK_TRY(dev_reserve(ctx, ctx->flags, 64 + (size_t)(n_tiles + 1) * 8 + (size_t)n_tiles * 4 + 64));
    const uint64_t *d_spare = ctx->flags.as<uint64_t>();      // 64 bytes the lanes without a list read (contents irrelevant)
    uint64_t *d_bounds = ctx->flags.as<uint64_t>() + 8;
    uint32_t *d_rows = reinterpret_cast<uint32_t *>(d_bounds + n_tiles + 1);
    PSK_TRY(dev_reserve(ctx, ctx->keysA, n_bmw * 8 + 64));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, (n_bmw + 1) * 4 + 64));
    unsigned long long *gbm = ctx->keysA.as<unsigned long long>();
    uint32_t *rank = ctx->keysB.as<uint32_t>();
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 2;
    PSK_HIP(ctx, hipMemcpyAsync(d_bounds, bounds.data(), (size_t)(n_tiles + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_groups > 1) PSK_HIP(ctx, hipMemsetAsync(gbm, 0, n_bmw * 8, ctx->stream));
    const dim3 grid((unsigned)n_ranges, (unsigned)n_groups);
    mark("bounds + buffers");
    // words relative to the slab's base fit 32 bits (every k <= 16) -- with 0xFFFFFFFF to spare: the 32-bit cursors use it as
    // "no word left".  The whole k = 16 space is safe (T...T is never canonical); a k = 17 slab cut at list quantiles whose
    // 64-aligned span is exactly 2^32 could hold a real word there: it takes the 64-bit cursors
    const bool w32 = span < (1ull << 32) || (span == (1ull << 32) && base == 0 && k == 16);
    const size_t ring_bytes = w32 ? (size_t)threads * PSK_PM_BLOCK * 2 * 4 : 0;   // the lanes' windows (32-bit words): 64 KB for 1,024 lanes
    if (w32 && ring_bytes + PM_BMW * 8 > 64 * 1024)
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pm_mark_kernel<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_bytes));
    // records of pass 1 for pass 2 (PmRec): a pool of pairs / PSK_MERGE_REC_DIV records (default 12; 0: none, pass 2 merges)
    PmRec rec = {nullptr, nullptr, nullptr, nullptr, 0};   // (region_chunks = 0: no records)
    {
        uint64_t div = 12;
        if (const char *e = getenv("PSK_MERGE_REC_DIV")) {
            char *end = nullptr;
            div = strtoull(e, &end, 10);
            if (!*e || *end) return psk_fail(ctx, PSK_EINVAL, "PSK_MERGE_REC_DIV=%s: expected a whole number (0 = no records)", e);
        }
        // The pool has a fixed part -- every wave's claims, whatever it holds -- of about a gigabyte: a build of a few million
        // pairs gains nothing from records (its merging pass 2 takes microseconds) and is spared that reservation, unless the
        // knob asks for records explicitly (tests of the replay on small sets)
        if (div && total_pairs < (16ull << 20) && !getenv("PSK_MERGE_REC_DIV")) div = 0;
        if (div) {
            // + every wave's claims under way (two) and the one it ends in; a region takes what the ranges that map to it need:
            // half as much again for their imbalance
            uint64_t chunks = total_pairs / div / 64 + 3 * PM_REC_BLOCK * (uint64_t)n_ranges * n_groups * (threads / 64);
            uint64_t region_chunks = ((chunks + chunks / 2) / PM_REC_REGIONS + PM_REC_BLOCK) & ~(uint64_t)(PM_REC_BLOCK - 1);
            if (const char *e = getenv("PSK_MERGE_REC_REGION")) {   // (tests: a region too small on purpose)
                const uint64_t v = strtoull(e, nullptr, 10);
                if (v >= 1 && v < (1ull << 26)) region_chunks = (v + PM_REC_BLOCK - 1) & ~(uint64_t)(PM_REC_BLOCK - 1);
            }
            chunks = region_chunks * PM_REC_REGIONS;
            if (chunks < (1ull << 31)) {
                const size_t wbytes = w32 ? 4 : 8, ctr_bytes = (size_t)(PM_REC_REGIONS + 1) * PM_REC_CTR_STRIDE * 4;
                // a pool that cannot be had (memory held by the lists and the matrix of a large run) costs the replay, not the
                // build: without records pass 2 merges again
                if (dev_reserve(ctx, ctx->valsA, chunks * 64 * (wbytes + 8)) == PSK_OK &&
                    dev_reserve(ctx, ctx->valsB, ctr_bytes + chunks * 8) == PSK_OK) {
                    rec.ctr = ctx->valsB.as<uint32_t>();
                    rec.hdr = reinterpret_cast<uint2 *>(ctx->valsB.as<uint8_t>() + ctr_bytes);
                    rec.masks = ctx->valsA.as<unsigned long long>();
                    rec.words = ctx->valsA.as<uint8_t>() + chunks * 64 * 8;
                    rec.region_chunks = (uint32_t)region_chunks;
                    PSK_HIP(ctx, hipMemsetAsync(ctx->valsB.p, 0, ctr_bytes + chunks * 8, ctx->stream));
                } else {
                    ctx->err.clear();
                    (void)hipGetLastError();
                }
            }
        }
    }
    if (w32) pm_mark_kernel<uint32_t><<<grid, threads, ring_bytes, ctx->stream>>>(d_refs, n, d_bounds, n_tiles, tiles_per_range, base, gbm, n_groups == 1, d_spare, rec);
    else pm_mark_kernel<uint64_t><<<grid, threads, 0, ctx->stream>>>(d_refs, n, d_bounds, n_tiles, tiles_per_range, base, gbm, n_groups == 1, d_spare, rec);
    PSK_HIP(ctx, hipGetLastError());
    mark("pm_mark");
    pm_popcount_kernel<<<div_up(n_bmw + 1, 256), 256, 0, ctx->stream>>>(gbm, n_bmw, rank);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(dev_exclusive_scan_u32(ctx, rank, rank, n_bmw + 1, d_m));
    pm_tile_rows_kernel<<<div_up(n_tiles, 256), 256, 0, ctx->stream>>>(rank, d_bounds, n_tiles, base, d_rows);
    PSK_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> rows(n_tiles);
    uint32_t m32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(rows.data(), d_rows, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(&m32, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    uint32_t rec_state[(PM_REC_REGIONS + 1) * PM_REC_CTR_STRIDE] = {0};   // chunks claimed per region, overflow
    if (rec.region_chunks) PSK_HIP(ctx, hipMemcpyAsync(rec_state, rec.ctr, sizeof(rec_state), hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `bounds` (host) has been copied as well
    mark("ranks");
    const uint64_t M = m32;
    uint32_t rows_max = 0;
    for (uint32_t t = 0; t < n_tiles; t++) rows_max = std::max(rows_max, rows[t]);
    PSK_TRY(dev_reserve(ctx, ctx->union_words, (M ? M : 1) * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, (M ? M : 1) * (uint64_t)wpr * 8));
    mark("alloc matrix");
    const bool replay = rec.region_chunks && !rec_state[PM_REC_REGIONS * PM_REC_CTR_STRIDE];
    if (M && replay) {
        uint32_t n_chunks = 0;   // the most a region has claimed
        for (int r = 0; r < PM_REC_REGIONS; r++) n_chunks = std::max(n_chunks, std::min(rec_state[r * PM_REC_CTR_STRIDE], rec.region_chunks));
        PSK_HIP(ctx, hipMemsetAsync(ctx->bits.p, 0, M * (uint64_t)wpr * 8, ctx->stream));
        const dim3 rgrid((unsigned)div_up(n_chunks, 4), PM_REC_REGIONS);
        if (w32) pm_replay_kernel<uint32_t><<<rgrid, 256, 0, ctx->stream>>>(rec, n_chunks, gbm, rank, wpr, ctx->bits.as<uint64_t>());
        else pm_replay_kernel<uint64_t><<<rgrid, 256, 0, ctx->stream>>>(rec, n_chunks, gbm, rank, wpr, ctx->bits.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        pm_union_kernel<<<div_up(n_bmw, 256), 256, 0, ctx->stream>>>(gbm, rank, n_bmw, base, ctx->union_words.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (trace) fprintf(stderr, "[psk]   merge records: at most %u chunks of 64 in a region (of %u), %d regions\n", n_chunks, rec.region_chunks, PM_REC_REGIONS);
    } else if (M) {
        if (trace && rec.region_chunks) fprintf(stderr, "[psk]   merge records: a region of %u chunks overflowed, pass 2 merges\n", rec.region_chunks);
        const int cols0 = wpr < PM_COLS ? wpr : PM_COLS;
        const size_t head = ring_bytes + (size_t)bmw_max * 8 + (size_t)((bmw_max + 2) >> 1) * 8;
        // rows a block holds: what fits beside the bitmap -- but sized for the bulk of the tiles (99.5th percentile of
        // their row counts, and <= 72 KB so that two workgroups share a CU), not for the one tile in a thousand that the
        // pilot cut too wide: those are streamed in several batches (first cut: the widest tile of config 3's slab, 2,461
        // rows against a mean of 152, made every workgroup reserve 147 KB of LDS: one workgroup per CU, 13.6 ms)
        uint32_t r_cap = (uint32_t)((PM_LDS_MAX - head) / ((size_t)cols0 * 8));
        {   // two workgroups per CU (78 KB each) whenever the bulk of the tiles -- the 99.5th percentile of their row counts, or
            // PSK_MERGE_RCAP_PCT -- fits a block of that size; the few wider tiles are streamed in batches
            double pct = 0.995;
            if (const char *pe = getenv("PSK_MERGE_RCAP_PCT")) { const double v = atof(pe); if (v > 0 && v <= 1) pct = v; }
            std::vector<uint32_t> sorted_rows(rows);
            const size_t q = (size_t)((double)(n_tiles - 1) * pct);
            std::nth_element(sorted_rows.begin(), sorted_rows.begin() + q, sorted_rows.end());
            const uint32_t bulk = sorted_rows[q] < 64 ? 64 : sorted_rows[q];
            const size_t two_per_cu = 78 * 1024;
            // (only a build whose fill kernel fits two workgroups per CU by its registers asks for it -- EXTRA="-DPSK_PM_BLOCK=4
            // -DPSK_PM_FILL_WGS=8": 57 VGPRs, a 32-KB ring; measured no faster than one workgroup with 64-byte blocks: fill 13.0
            // against 13.6 ms, mark 6.2 against 5.3)
            if ((PSK_PM_FILL_WGS >= 8 || getenv("PSK_MERGE_RCAP_PCT")) && head + (size_t)cols0 * 8 * bulk <= two_per_cu)
                r_cap = std::min<uint32_t>(r_cap, (uint32_t)((two_per_cu - head) / ((size_t)cols0 * 8)));
        }
        const uint32_t rb_max = rows_max < r_cap ? rows_max : r_cap;
        const size_t lds = head + (size_t)cols0 * rb_max * 8;
        auto fill = w32 ? pm_fill_kernel<uint32_t> : pm_fill_kernel<uint64_t>;
        if (lds > 64 * 1024)
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fill<<<grid, threads, lds, ctx->stream>>>(d_refs, n, wpr, d_bounds, n_tiles, tiles_per_range, base, gbm, rank, r_cap, bmw_max,
                                                  ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>(), d_spare);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark("pm_fill");
#ifdef PSK_PM_STATS
    {
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(pm_stats), sizeof(h));
        const double w = (double)h[7];   // waves
        fprintf(stderr, "[psk] pm_fill per wave (us of 100 MHz ticks / 100): loads+sync %.1f, zero+sync %.1f, merge %.1f, wait %.1f, copy-out %.1f, union+sync %.1f; iterations %.0f per wave\n",
                h[0] / w / 100, h[1] / w / 100, h[2] / w / 100, h[3] / w / 100, h[4] / w / 100, h[5] / w / 100, h[6] / w);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pm_stats), z, sizeof(z));
    }
#endif
    if (getenv("PSK_TRACE"))
        fprintf(stderr, "[psk] merge build: %u tiles (%llu pairs each wanted), %llu ranges x %d groups of %d threads, widest tile %u bitmap words, most rows %u\n",
                n_tiles, (unsigned long long)pairs_per_tile, (unsigned long long)n_ranges, n_groups, threads, bmw_max, rows_max);
    *n_kmers = M;
    *done = 1;
    return PSK_OK;
}

// ---- r05: the merge build without a value bitmap (word spaces beyond 2^34 values: k >= 18) -------------------------------------
// The same streaming 64-way merge per wave (pm_mark_kernel<uint64_t, false>) leaves its (word, ballot) records; the union is the
// sorted distinct record words (one radix sort of the RECORD words -- a twentieth of the pairs at config 3's sharing, against the
// sort route's sort of every (word, sample) pair), a record's row is its word's place in the union (a cell table + a short
// search), and pass 2 is the replay: one 8-byte store per record into the zeroed matrix.  When the samples share too little for
// the record pool (records ~ pairs) the build declines and the sort route takes over.
// Returns PSK_OK and sets *done = 1 when it ran.
int build_presence_merge_wide(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done)
{
    *done = 0;
    if (getenv("PSK_NO_MERGE_PRESENCE") || getenv("PSK_NO_WIDE_MERGE")) return PSK_OK;
    const bool trace = getenv("PSK_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[psk]   wide merge %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const int k = ctx->k, n = ctx->n_samples, wpr = ctx->wpr;
    if (k < 1 || k > 32 || n < 1 || total_pairs == 0) return PSK_OK;
    for (int i = 0; i < n; i++)
        if (ctx->lists[i].n_unique >= (1ull << 32)) return PSK_OK;
    const uint64_t lo = ctx->slab_lo;
    const uint64_t hi = ctx->slab_hi ? ctx->slab_hi : (k == 32 ? ~0ull : (1ull << (2 * k)));   // exclusive; no canonical word is all ones
    if (hi <= lo) return PSK_OK;
    // ---- tile bounds: pair quantiles of a pilot (they only cut the stream into ranges of work: no bitmap depends on them) ----
    uint64_t pairs_per_tile = 8192;
    if (const char *e = getenv("PSK_MERGE_TILE_PAIRS")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 64) pairs_per_tile = v; }
    uint64_t want = (total_pairs + pairs_per_tile - 1) / pairs_per_tile;
    if (want < 1) want = 1;
    if (want > (1ull << 24)) want = 1ull << 24;
    const int n_pick = n < 64 ? n : 64;
    std::vector<int32_t> pick(n_pick);
    for (int i = 0; i < n_pick; i++) pick[i] = (int32_t)(((int64_t)i * n) / n_pick);
    uint64_t per_list = (4 * want + n_pick - 1) / n_pick;
    if (per_list < 64) per_list = 64;
    if (per_list > (1u << 22)) per_list = 1u << 22;
    const uint64_t n_pilot = per_list * (uint64_t)n_pick;
    std::vector<PmList> refs(n);
    for (int i = 0; i < n; i++) { refs[i].words = ctx->lists[i].words; refs[i].n = ctx->lists[i].n_unique; }
    PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)n * sizeof(PmList) + (size_t)n_pick * 4 + 64));
    PmList *d_refs = ctx->starts.as<PmList>();
    int32_t *d_pick = reinterpret_cast<int32_t *>(d_refs + n);
    PSK_HIP(ctx, hipMemcpyAsync(d_refs, refs.data(), (size_t)n * sizeof(PmList), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d_pick, pick.data(), (size_t)n_pick * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->keysA, n_pilot * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, n_pilot * 8));
    pm_pilot_kernel<<<div_up(n_pilot, 256), 256, 0, ctx->stream>>>(d_refs, d_pick, n_pick, (uint32_t)per_list, ctx->keysA.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    uint64_t *sorted = nullptr;
    // (the sentinels of empty lists sort last on the low 2k bits alone: all ones there is no canonical word)
    PSK_TRY(dev_radix_sort_u64(ctx, ctx->keysA.as<uint64_t>(), ctx->keysB.as<uint64_t>(), n_pilot, 0, 2 * k, &sorted));
    std::vector<uint64_t> pilot(n_pilot);
    PSK_HIP(ctx, hipMemcpyAsync(pilot.data(), sorted, n_pilot * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mark("pilot + sort");
    while (!pilot.empty() && pilot.back() == PM_SENT) pilot.pop_back();
    std::vector<uint64_t> bounds;
    bounds.push_back(lo);
    if (!pilot.empty())
        for (uint64_t j = 1; j < want; j++) {
            const uint64_t v = pilot[(size_t)((pilot.size() * j) / want)];
            if (v > bounds.back() && v < hi) bounds.push_back(v);
        }
    bounds.push_back(hi);
    uint32_t n_tiles = (uint32_t)(bounds.size() - 1);
    // launch shape: groups of `gsz` samples (one workgroup each), ranges of consecutive tiles
    int n_groups = 0, threads = 0;
    uint64_t n_ranges = 0;
    uint32_t tiles_per_range = 0;
    auto shape = [&](int gsz) {
        n_groups = (n + gsz - 1) / gsz;
        threads = n >= gsz ? gsz : ((n + 63) / 64) * 64;
        n_ranges = ((uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8 * (PM_GROUP / threads)) / n_groups;
        if (const char *e = getenv("PSK_MERGE_RANGES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1) n_ranges = v; }
        if (n_ranges > n_tiles) n_ranges = n_tiles;
        if (n_ranges < 1) n_ranges = 1;
        tiles_per_range = (uint32_t)((n_tiles + n_ranges - 1) / n_ranges);
        n_ranges = (n_tiles + tiles_per_range - 1) / tiles_per_range;
    };
    shape(PM_GROUP);
    // 32-bit cursors (words relative to the range's first bound; the inline-asm DPP minimum, a 64-KB ring) when every range of the
    // launch spans less than 2^32 - 1 word values; else 64-bit ones, in groups of 512 samples: their ring is 128 bytes per lane, and
    // 64 KB of it per workgroup leave room for two workgroups on a CU.  Tiles are cut at pair quantiles of a pilot, so a few come out
    // many times wider than the rest: a tile wider than its share of a range's 2^32 is cut further, by value (a bound may be any
    // word value) -- unless that would more than double the tiles (a space as sparse as k = 31's: the 64-bit cursors are for it)
    bool narrow = !getenv("PSK_WIDE_MERGE_64");
    const uint64_t limit = 0xfffffffdull;
    uint64_t widest = 0;
    for (int round = 0; narrow && round < 4; round++) {
        const uint64_t cap = limit / tiles_per_range;
        uint64_t extra = 0;
        for (uint32_t t = 0; t < n_tiles; t++) extra += (bounds[t + 1] - bounds[t] - 1) / cap;
        if (extra == 0) break;
        if (extra > n_tiles) { narrow = false; break; }
        std::vector<uint64_t> cut;
        cut.reserve(bounds.size() + extra);
        for (uint32_t t = 0; t < n_tiles; t++) {
            const uint64_t a0 = bounds[t], span = bounds[t + 1] - a0, parts = (span - 1) / cap + 1;
            for (uint64_t q = 0; q < parts; q++) cut.push_back(a0 + (span / parts) * q + std::min<uint64_t>(q, span % parts));
        }
        cut.push_back(bounds[n_tiles]);
        bounds.swap(cut);
        n_tiles = (uint32_t)(bounds.size() - 1);
        shape(PM_GROUP);
    }
    for (uint64_t r0 = 0; r0 < n_tiles; r0 += tiles_per_range) {
        const uint64_t r1 = std::min<uint64_t>(r0 + tiles_per_range, n_tiles);
        widest = std::max(widest, bounds[r1] - bounds[r0]);
    }
    if (widest > limit) narrow = false;
    if (trace) fprintf(stderr, "[psk]   wide merge: %u tiles, %u per range, widest range %llu word values\n", n_tiles, tiles_per_range, (unsigned long long)widest);
    if (!narrow) {
        int gsz = 512;
        if (const char *e = getenv("PSK_WIDE_GROUP")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) gsz = v; }
        shape(gsz);
    }
    PSK_TRY(dev_reserve(ctx, ctx->flags, 64 + (size_t)(n_tiles + 1) * 8 + 64));
    const uint64_t *d_spare = ctx->flags.as<uint64_t>();
    uint64_t *d_bounds = ctx->flags.as<uint64_t>() + 8;
    PSK_HIP(ctx, hipMemcpyAsync(d_bounds, bounds.data(), (size_t)(n_tiles + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    // ---- the record pool: pairs / PSK_MERGE_REC_DIV records (default 12) + every wave's claims under way ------------------------
    // (a record stands for the samples of one wave that share a word: a few dozen samples cannot share twelve-fold)
    uint64_t div = std::min<uint64_t>(12, std::max<uint64_t>(2, (uint64_t)n / 8));
    if (const char *e = getenv("PSK_MERGE_REC_DIV")) {
        char *end = nullptr;
        div = strtoull(e, &end, 10);
        if (!*e || *end) return psk_fail(ctx, PSK_EINVAL, "PSK_MERGE_REC_DIV=%s: expected a whole number (0 = no records)", e);
    }
    if (div == 0) return PSK_OK;   // no records, no wide merge: the sort route
    uint64_t chunks = total_pairs / div / 64 + 3 * PM_REC_BLOCK * (uint64_t)n_ranges * n_groups * (threads / 64);
    // a workgroup's records all go to ONE region ((range + group) mod 16): with fewer workgroups than regions only that many
    // regions are ever used, and here -- unlike in the bitmap build, whose pass 2 can merge again -- an overflow costs the route
    const uint64_t regions_used = std::min<uint64_t>(PM_REC_REGIONS, n_ranges * (uint64_t)n_groups);
    uint64_t region_chunks = ((chunks + chunks / 2) / regions_used + PM_REC_BLOCK) & ~(uint64_t)(PM_REC_BLOCK - 1);
    if (const char *e = getenv("PSK_MERGE_REC_REGION")) {
        const uint64_t v = strtoull(e, nullptr, 10);
        if (v >= 1 && v < (1ull << 26)) region_chunks = (v + PM_REC_BLOCK - 1) & ~(uint64_t)(PM_REC_BLOCK - 1);
    }
    chunks = region_chunks * PM_REC_REGIONS;
    if (chunks >= (1ull << 26)) return PSK_OK;          // records are numbered in u32: 64 x chunks < 2^32
    const size_t ctr_bytes = (size_t)(PM_REC_REGIONS + 1) * PM_REC_CTR_STRIDE * 4;
    if (dev_reserve(ctx, ctx->valsA, chunks * 64 * 16) != PSK_OK || dev_reserve(ctx, ctx->valsB, ctr_bytes + chunks * 8) != PSK_OK) {
        ctx->err.clear();
        (void)hipGetLastError();
        return PSK_OK;
    }
    PmRec rec;
    rec.ctr = ctx->valsB.as<uint32_t>();
    rec.hdr = reinterpret_cast<uint2 *>(ctx->valsB.as<uint8_t>() + ctr_bytes);
    rec.masks = ctx->valsA.as<unsigned long long>();
    rec.words = ctx->valsA.as<uint8_t>() + chunks * 64 * 8;
    rec.region_chunks = (uint32_t)region_chunks;
    PSK_HIP(ctx, hipMemsetAsync(ctx->valsB.p, 0, ctr_bytes + chunks * 8, ctx->stream));
    mark("bounds + buffers");
    const dim3 grid((unsigned)n_ranges, (unsigned)n_groups);
    const size_t ring_bytes = (size_t)threads * PSK_PM_BLOCK * 2 * (narrow ? 4 : 8);   // the lanes' windows: 64 / 128 KB for 1,024 lanes
    auto mark_k = narrow ? pm_mark_kernel<uint32_t, false> : pm_mark_kernel<uint64_t, false>;
    if (ring_bytes > 48 * 1024)
        PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(mark_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_bytes));
    mark_k<<<grid, threads, ring_bytes, ctx->stream>>>(d_refs, n, d_bounds, n_tiles, tiles_per_range, 0ull, nullptr, 0, d_spare, rec);
    PSK_HIP(ctx, hipGetLastError());
    uint32_t rec_state[(PM_REC_REGIONS + 1) * PM_REC_CTR_STRIDE] = {0};
    PSK_HIP(ctx, hipMemcpyAsync(rec_state, rec.ctr, sizeof(rec_state), hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mark("pm_mark (records)");
    if (rec_state[PM_REC_REGIONS * PM_REC_CTR_STRIDE]) {
        if (trace) fprintf(stderr, "[psk]   wide merge: a record region of %u chunks overflowed (the samples share too little): the sort route\n", rec.region_chunks);
        return PSK_OK;
    }
    // ---- the record words, dense: chunk counts -> offsets -> keys ------------------------------------------------------------
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_tot = ctx->misc.as<uint32_t>() + 2;
    uint32_t n_rel = 0;      // the most chunks a region has claimed
    for (int r = 0; r < PM_REC_REGIONS; r++) n_rel = std::max(n_rel, std::min(rec_state[r * PM_REC_CTR_STRIDE], rec.region_chunks));
    const uint64_t n_dense = (uint64_t)n_rel * PM_REC_REGIONS;
    const size_t cells_at = ((size_t)(n_dense + 2) * 4 + 255) & ~(size_t)255;
    PSK_TRY(dev_reserve(ctx, ctx->raw, cells_at + (((size_t)1 << 22) + 2) * 4));   // chunk offsets | the cell table of the replay
    uint32_t *d_off = ctx->raw.as<uint32_t>();
    pmw_chunk_counts_kernel<<<div_up(n_dense + 1, 256), 256, 0, ctx->stream>>>(rec.hdr, rec.region_chunks, n_rel, d_off);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(dev_exclusive_scan_u32(ctx, d_off, d_off, n_dense + 1, d_tot));
    uint32_t n_rec = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&n_rec, d_tot, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->keysA, ((size_t)n_rec + 1) * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, ((size_t)n_rec + 1) * 8));
    if (n_rel)
        pmw_gather_kernel<<<dim3(div_up(n_rel, 4), PM_REC_REGIONS), 256, 0, ctx->stream>>>(rec.hdr, reinterpret_cast<const uint64_t *>(rec.words),
                                                                                         rec.region_chunks, n_rel, d_off, lo, ctx->keysA.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    mark("record words");
    uint64_t *keys = ctx->keysA.as<uint64_t>();
    int key_bits = 1;
    while (key_bits < 64 && ((hi - 1 - lo) >> key_bits) != 0) key_bits++;
    if (n_rec) PSK_TRY(dev_radix_sort_u64(ctx, ctx->keysA.as<uint64_t>(), ctx->keysB.as<uint64_t>(), n_rec, 0, key_bits, &keys));
    mark("sort of the words");
    // ---- distinct words = the union; their places = the rows ------------------------------------------------------------------
    PSK_TRY(dev_reserve(ctx, ctx->hist, ((size_t)n_rec + 2) * 4));   // (the sort, which uses this buffer, is over)
    uint32_t *d_head = ctx->hist.as<uint32_t>();
    pmw_heads_kernel<<<div_up((uint64_t)n_rec + 1, 256), 256, 0, ctx->stream>>>(keys, n_rec, d_head);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(dev_exclusive_scan_u32(ctx, d_head, d_head, (uint64_t)n_rec + 1, d_tot));
    uint32_t m32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&m32, d_tot, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t M = m32;
    mark("heads + ranks");
    PSK_TRY(dev_reserve(ctx, ctx->union_words, (M ? M : 1) * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, (M ? M : 1) * (uint64_t)wpr * 8));
    mark("alloc matrix");
    if (M) {
        pmw_union_kernel<<<div_up(n_rec, 256), 256, 0, ctx->stream>>>(keys, n_rec, d_head, lo, ctx->union_words.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        // cells over [lo, hi): about four rows each, at most 2^22 of them
        uint32_t n_cells = 1024;
        while (n_cells < (1u << 22) && (uint64_t)n_cells * 4 < M) n_cells <<= 1;
        uint32_t shift = 0;
        while (((hi - 1 - lo) >> shift) >= (uint64_t)n_cells) shift++;
        uint32_t *d_cells = reinterpret_cast<uint32_t *>(ctx->raw.as<uint8_t>() + cells_at);
        pmw_cells_kernel<<<div_up(n_cells + 1, 256), 256, 0, ctx->stream>>>(ctx->union_words.as<uint64_t>(), (uint32_t)M, lo, shift, n_cells, d_cells);
        PSK_HIP(ctx, hipGetLastError());
        mark("union + cells");
        PSK_HIP(ctx, hipMemsetAsync(ctx->bits.p, 0, M * (uint64_t)wpr * 8, ctx->stream));
        pmw_replay_kernel<<<dim3(div_up(n_rel, 4), PM_REC_REGIONS), 256, 0, ctx->stream>>>(rec, n_rel, ctx->union_words.as<uint64_t>(), d_cells, lo, shift,
                                                                                         n_cells, wpr, ctx->bits.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        mark("memset + replay");
    }
    if (trace)
        fprintf(stderr, "[psk] wide merge build: %u tiles, %llu ranges x %d groups of %d threads (%d-bit cursors), %u records for %llu pairs, %llu rows\n", n_tiles,
                (unsigned long long)n_ranges, n_groups, threads, narrow ? 32 : 64, n_rec, (unsigned long long)total_pairs, (unsigned long long)M);
    *n_kmers = M;
    *done = 1;
    return PSK_OK;
}
