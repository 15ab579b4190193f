// a1: per-sample canonical k-mer list (replaces bin/glistmaker; modeling.py:303-315), the
// count lookups of modeling.py:324-329 and the fixed-dictionary counting of prediction.py:72-80.
//
//   host   frame_sequence_host   record framing only (headers, FASTQ separator/quality lines,
//                                control bytes) -> clean stream: bases + '\n' window breaks
//   device extract_kernel        2-bit encode, rolling forward/reverse words, canonical min,
//                                slab filter, per-wave LDS compaction + one reservation per wave
//          dev_radix_sort_u64    LSD radix sort on the 2k significant bits
//          rle_* kernels         run heads -> unique words + u32 frequencies
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "dev_utils.h"
#include "psk_internal.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

// ------------------------------------------------------------------------------------------------
// Host framing.  Tokeniser contract of glistmaker 4.2.3 as established by probing the binary
// (DESIGN.md "Tokeniser contract"; fixtures tests/golden/tokenizer_cases.json).
// ------------------------------------------------------------------------------------------------
namespace {
enum { ST_INIT, ST_FA_HDR, ST_FA_SEQ, ST_FQ_HDR, ST_FQ_SEQ, ST_FQ_PLUS, ST_FQ_QUAL, ST_FQ_H, ST_FQ_HSKIP };
enum { CL_BREAK = 0, CL_BASE = 1, CL_SKIP = 2 };

struct ClassTable {
    uint8_t t[256];
    ClassTable()
    {
        for (int c = 0; c < 256; c++) t[c] = (c < 32) ? CL_SKIP : CL_BREAK;
        for (const char *p = "ACGTUacgtu"; *p; p++) t[(unsigned char)*p] = CL_BASE;
    }
};
const ClassTable g_cls;

static inline bool is_base_byte(uint8_t c)
{
    const uint8_t x = c | 0x20;  // fold case
    return (x == 'a') | (x == 'c') | (x == 'g') | (x == 't') | (x == 'u');
}

// length of the leading run of base bytes of p[0..n): 16 bytes per step on the host's SSE2 unit
static inline size_t base_run_length(const uint8_t *p, size_t n)
{
    size_t j = 0;
#if defined(__SSE2__)
    const __m128i fold = _mm_set1_epi8(0x20);
    const __m128i ca = _mm_set1_epi8('a'), cc = _mm_set1_epi8('c'), cg = _mm_set1_epi8('g'), ct = _mm_set1_epi8('t'),
                  cu = _mm_set1_epi8('u');
    while (j + 16 <= n) {
        const __m128i v = _mm_or_si128(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + j)), fold);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, ca), _mm_cmpeq_epi8(v, cc)),
                                        _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, cg), _mm_cmpeq_epi8(v, ct)),
                                                     _mm_cmpeq_epi8(v, cu)));
        const unsigned m = (unsigned)_mm_movemask_epi8(ok);
        if (m != 0xffffu) return j + (size_t)__builtin_ctz(~m);
        j += 16;
    }
#endif
    while (j < n && is_base_byte(p[j])) j++;
    return j;
}
}  // namespace

// k > 0: *n_windows receives the number of k-base windows of the clean stream (sum over its runs of
// max(0, run - k + 1)), i.e. the number of words the extract kernel emits when no slab filter is set.
static int64_t frame_sequence_counting(const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap, int k,
                                       uint64_t *n_windows)
{
    size_t o = 0;
    int st = ST_INIT;
    bool last_break = true;  // collapse runs of breaks; no leading break needed
    size_t run_start = 0;    // output offset where the current run of bases began
    uint64_t wins = 0;
    auto close_run = [&]() {
        const size_t run = o - run_start;
        if (k > 0 && run >= (size_t)k) wins += run - (size_t)k + 1;
    };
    auto put_break = [&]() {
        if (!last_break) { close_run(); out[o++] = '\n'; last_break = true; run_start = o; }
    };
    auto finish = [&]() -> int64_t {
        if (!last_break) close_run();
        if (n_windows) *n_windows = wins;
        return (int64_t)o;
    };
    if (out_cap < len) return PSK_ERANGE;
    for (size_t i = 0; i < len; i++) {
        const uint8_t c = bytes[i];
        if (c == 0) break;
        if ((st == ST_FA_HDR || st == ST_FQ_HDR || st == ST_FQ_PLUS || st == ST_FQ_QUAL || st == ST_FQ_HSKIP) && c != '\n') {
            // skip to the end of this line in one go
            const void *nl = memchr(bytes + i, '\n', len - i);
            const size_t j = nl ? (size_t)(static_cast<const uint8_t *>(nl) - bytes) : len;
            if (memchr(bytes + i, 0, j - i)) break;  // a NUL inside the skipped text ends the input
            if (j >= len) break;
            i = j - 1;  // the newline itself goes through the state machine
            continue;
        }
        switch (st) {
        case ST_INIT:
            if (c == '>') st = ST_FA_HDR;
            else if (c == '@') st = ST_FQ_HDR;
            break;
        case ST_FA_HDR:
            if (c == '\n') { st = ST_FA_SEQ; put_break(); }
            break;
        case ST_FQ_HDR:
            if (c == '\n') { st = ST_FQ_SEQ; put_break(); }
            break;
        case ST_FA_SEQ:
        case ST_FQ_SEQ: {
            // fast path: a run of base bytes is copied in one go (vectorisable scan, no table look-up)
            if (is_base_byte(c)) {
                const size_t j = i + base_run_length(bytes + i, len - i);
                memcpy(out + o, bytes + i, j - i);
                o += j - i;
                last_break = false;
                i = j - 1;
                break;
            }
            const uint8_t cl = g_cls.t[c];
            if (cl == CL_BASE) {
                out[o++] = c;
                last_break = false;
            } else if (st == ST_FA_SEQ && c == '>') {
                put_break();
                st = ST_FA_HDR;
            } else if (cl == CL_SKIP) {
                if (st == ST_FQ_SEQ && c == '\n' && i + 1 < len) {
                    const uint8_t c2 = bytes[++i];  // the byte after a sequence newline is consumed
                    if (c2 == 0) return finish();
                    if (c2 == '+') st = ST_FQ_PLUS;
                }
            } else {
                put_break();
            }
            break;
        }
        case ST_FQ_PLUS:
            if (c == '\n') st = ST_FQ_QUAL;
            break;
        case ST_FQ_QUAL:
            if (c == '\n') st = ST_FQ_H;
            break;
        case ST_FQ_H:
            if (c == '@') { st = ST_FQ_HDR; put_break(); }
            else st = ST_FQ_HSKIP;
            break;
        case ST_FQ_HSKIP:
            if (c == '\n') st = ST_FQ_H;
            break;
        }
    }
    return finish();
}

int64_t frame_sequence_host(const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap)
{
    return frame_sequence_counting(bytes, len, out, out_cap, 0, nullptr);
}

extern "C" int64_t psk_frame_sequence(const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap)
{
    if ((!bytes && len) || !out) return PSK_EINVAL;
    return frame_sequence_host(bytes, len, out, out_cap);
}

// ------------------------------------------------------------------------------------------------
// Device kernels
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int EX_THREADS = 256;
constexpr int EX_SEG = 32;   // window-end positions per thread
constexpr int EX_HALO = 32;  // bytes before the segment that are rolled first (k - 1 <= 31)

struct Roll {
    uint64_t fw, rc;
    int run;
};

__device__ __forceinline__ void roll_byte(Roll &r, uint32_t c, uint64_t mask, int rcshift, int k)
{
    if (c == '\n') {
        r.run = 0;
    } else {
        const uint64_t code = ((c >> 1) ^ (c >> 2)) & 3u;  // A/a 0, C/c 1, G/g 2, T/t/U/u 3
        r.fw = ((r.fw << 2) | code) & mask;
        r.rc = (r.rc >> 2) | ((3ull - code) << rcshift);
        r.run = (r.run < k) ? r.run + 1 : k;
    }
}

// clean: bases and '\n' breaks, 16-byte aligned, padded with '\n' to a multiple of EX_SEG.
// Every lane rolls EX_SEG consecutive window ends; the wave compacts its valid words into its own
// 16 KiB LDS region (ballot ranks, wave-uniform running count); the workgroup reserves its output range
// with ONE atomic and every wave spills its region with consecutive lanes on consecutive addresses.
__global__ __launch_bounds__(EX_THREADS) void extract_kernel(const uint8_t *__restrict__ clean, uint64_t len, int k,
                                                              uint64_t lo, uint64_t hi, uint64_t *__restrict__ out,
                                                              uint32_t *__restrict__ n_out)
{
    __shared__ uint64_t stage[EX_THREADS / 64][64 * EX_SEG];
    const uint64_t g = (uint64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    const uint64_t s = g * EX_SEG;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const int rcshift = 2 * (k - 1);
    const bool active = s < len;

    uint32_t cur[EX_SEG / 4], prev[EX_HALO / 4];
#pragma unroll
    for (int j = 0; j < EX_SEG / 4; j++) cur[j] = 0x0a0a0a0au;
#pragma unroll
    for (int j = 0; j < EX_HALO / 4; j++) prev[j] = 0x0a0a0a0au;
    if (active) {
        const uint4 *p = reinterpret_cast<const uint4 *>(clean + s);
#pragma unroll
        for (int q = 0; q < EX_SEG / 16; q++) {
            const uint4 a = p[q];
            cur[4 * q] = a.x; cur[4 * q + 1] = a.y; cur[4 * q + 2] = a.z; cur[4 * q + 3] = a.w;
        }
        // the EX_HALO bytes before s, 16 at a time (what lies before the buffer counts as a break)
#pragma unroll
        for (int q = 0; q < EX_HALO / 16; q++) {
            const uint64_t back = (uint64_t)(EX_HALO / 16 - q) * 16;
            if (s >= back) {
                const uint4 c = *reinterpret_cast<const uint4 *>(clean + s - back);
                prev[4 * q] = c.x; prev[4 * q + 1] = c.y; prev[4 * q + 2] = c.z; prev[4 * q + 3] = c.w;
            }
        }
    }
    Roll r{0, 0, 0};
    // warm-up over the k-1 bytes before s (k-1 <= 31 < EX_HALO)
#pragma unroll
    for (int j = 0; j < EX_HALO; j++) {
        if (j >= EX_HALO - (k - 1)) {
            const uint32_t c = (prev[j >> 2] >> ((j & 3) * 8)) & 0xffu;
            roll_byte(r, c, mask, rcshift, k);
        }
    }
    uint32_t wcount = 0;  // wave-uniform
#pragma unroll
    for (int j = 0; j < EX_SEG; j++) {
        const uint32_t c = (cur[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        roll_byte(r, c, mask, rcshift, k);
        const uint64_t w = (r.fw < r.rc) ? r.fw : r.rc;
        const bool valid = active && (s + j < len) && (r.run >= k) && (w >= lo) && (hi == 0 || w < hi);
        const uint64_t bal = __ballot(valid);
        if (valid) stage[wid][wcount + __popcll(bal & psk_lanemask_lt(lane))] = w;
        wcount += (uint32_t)__popcll(bal);
    }
    // ONE reservation per workgroup: same-address atomics retire at ~11 ns each, so one per wave (2441 for a
    // 5-Mbp sample) was 27 of the kernel's 45 us
    __shared__ uint32_t s_wcount[EX_THREADS / 64];
    __shared__ uint32_t s_base;
    if (lane == 0) s_wcount[wid] = wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < EX_THREADS / 64; w++) tot += s_wcount[w];
        s_base = tot ? atomicAdd(n_out, tot) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (int w = 0; w < wid; w++) base += s_wcount[w];
    for (uint32_t i = lane; i < wcount; i += 64) out[(uint64_t)base + i] = stage[wid][i];
}

// ---- run-length encoding of the sorted words: unique words + u32 counts ---------------------------
// Two passes over the sorted keys and one tiny scan instead of flags / scan / scatter / counts:
//   rle_tile_kernel   per tile of 4096 keys: number of run heads, position of the first head
//   rle_tile_scan     one workgroup: exclusive scan of the head counts (-> output offset of every tile, total =
//                     number of unique words) and a suffix minimum of the first-head positions (-> where the
//                     run that is open at the end of a tile ends)
//   rle_emit_kernel   per tile again: head h writes its word and (position of the next head - its position)
// A wave covers 16 rows of 64 consecutive keys; the heads of a row are one ballot, so ranks and "next head"
// positions are scalar bit operations on wave-uniform masks.
constexpr int RLE_THREADS = 256;
constexpr int RLE_ROWS = 16;
constexpr int RLE_WAVE_KEYS = 64 * RLE_ROWS;                    // 1024
constexpr int RLE_TILE = RLE_WAVE_KEYS * (RLE_THREADS / 64);    // 4096

__device__ __forceinline__ uint64_t rle_row_heads(const uint64_t *__restrict__ keys, uint64_t n, uint64_t i, uint64_t *key_out)
{
    bool head = false;
    uint64_t k = 0;
    if (i < n) {
        k = keys[i];
        head = (i == 0) || keys[i - 1] != k;
    }
    *key_out = k;
    return __ballot(head);
}

// n_dev != nullptr: the key count sits in device memory (<= n_host, which sizes the grid)
__global__ __launch_bounds__(RLE_THREADS) void rle_tile_kernel(const uint64_t *__restrict__ keys, uint64_t n_host,
                                                               const uint32_t *__restrict__ n_dev,
                                                               uint32_t *__restrict__ tile_cnt,
                                                               uint32_t *__restrict__ tile_first)
{
    const uint64_t n = n_dev ? (uint64_t)*n_dev : n_host;
    __shared__ uint32_t s_cnt[RLE_THREADS / 64], s_first[RLE_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * RLE_TILE + (uint64_t)wid * RLE_WAVE_KEYS;
    uint32_t cnt = 0, first = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < RLE_ROWS; r++) {
        uint64_t k;
        const uint64_t m = rle_row_heads(keys, n, base + (uint64_t)r * 64 + lane, &k);
        if (m && first == 0xffffffffu) first = (uint32_t)(base + (uint64_t)r * 64 + __builtin_ctzll(m));
        cnt += __popcll(m);
    }
    if (lane == 0) { s_cnt[wid] = cnt; s_first[wid] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0, f = 0xffffffffu;
        for (int w = 0; w < RLE_THREADS / 64; w++) { c += s_cnt[w]; if (s_first[w] < f) f = s_first[w]; }
        tile_cnt[blockIdx.x] = c;
        tile_first[blockIdx.x] = f;
    }
}

// tile_cnt -> exclusive offsets (in place), tile_first -> position of the first head AFTER the tile (in place),
// total[0] = number of heads
__global__ __launch_bounds__(1024) void rle_tile_scan_kernel(uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ tile_first,
                                                              uint32_t n_tiles, uint32_t n_host, const uint32_t *__restrict__ n_dev,
                                                              uint32_t *__restrict__ total)
{
    const uint32_t n = n_dev ? *n_dev : n_host;
    __shared__ uint32_t lds[16];
    __shared__ uint32_t s_min[16];
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? tile_cnt[t] : 0u;
        uint32_t all;
        const uint32_t ex = psk_block_excl_scan_u32<1024>(v, &all, lds);
        if (t < n_tiles) tile_cnt[t] = carry + ex;
        carry += all;
    }
    if (threadIdx.x == 0) total[0] = carry;
    // suffix minimum, exclusive: next[t] = min(first[t+1 ..]) or n; chunks from the back
    uint32_t tail = n;  // minimum over everything behind the current chunk
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (uint32_t done = 0; done < n_tiles; done += 1024) {
        const uint32_t hi = n_tiles - done;                 // chunk = [lo, hi)
        const uint32_t lo = hi > 1024 ? hi - 1024 : 0;
        const uint32_t t = lo + threadIdx.x;
        const uint32_t v = t < hi ? tile_first[t] : 0xffffffffu;
        // inclusive suffix min within the wave (towards higher lanes), then across waves
        uint32_t m = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_down(m, d, 64);
            if (lane + d < 64 && o < m) m = o;
        }
        __syncthreads();
        if (lane == 0) s_min[wid] = m;  // minimum of the whole wave
        __syncthreads();
        uint32_t behind = tail;          // minimum of the waves behind this one + earlier chunks
        for (int w = wid + 1; w < 16; w++) if (s_min[w] < behind) behind = s_min[w];
        // exclusive: the inclusive suffix min of the next lane (or `behind` for the last lane)
        uint32_t nxt = __shfl_down(m, 1, 64);
        if (lane == 63) nxt = 0xffffffffu;
        uint32_t res = nxt < behind ? nxt : behind;
        uint32_t chunk_min = tail;
        for (int w = 0; w < 16; w++) if (s_min[w] < chunk_min) chunk_min = s_min[w];
        if (t < hi) tile_first[t] = res;
        tail = chunk_min;
        __syncthreads();
    }
}

__global__ __launch_bounds__(RLE_THREADS) void rle_emit_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                               const uint32_t *__restrict__ tile_off,
                                                               const uint32_t *__restrict__ tile_next,
                                                               uint64_t *__restrict__ words, uint32_t *__restrict__ freqs)
{
    __shared__ uint32_t s_cnt[RLE_THREADS / 64], s_first[RLE_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * RLE_TILE + (uint64_t)wid * RLE_WAVE_KEYS;
    uint64_t mask[RLE_ROWS], key[RLE_ROWS];
    uint32_t cnt = 0, first = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < RLE_ROWS; r++) {
        mask[r] = rle_row_heads(keys, n, base + (uint64_t)r * 64 + lane, &key[r]);
        if (mask[r] && first == 0xffffffffu) first = (uint32_t)(base + (uint64_t)r * 64 + __builtin_ctzll(mask[r]));
        cnt += __popcll(mask[r]);
    }
    if (lane == 0) { s_cnt[wid] = cnt; s_first[wid] = first; }
    __syncthreads();
    uint32_t out = tile_off[blockIdx.x];
    for (int w = 0; w < wid; w++) out += s_cnt[w];
    // first head behind this wave: a later wave of the tile, else the first head after the tile
    uint32_t after = tile_next[blockIdx.x];
    for (int w = RLE_THREADS / 64 - 1; w > wid; w--) if (s_first[w] != 0xffffffffu) after = s_first[w];
    // position of the first head in a later row of this wave, per row (scalar, back to front)
    uint32_t later[RLE_ROWS];
    uint32_t nxt = after;
#pragma unroll
    for (int r = RLE_ROWS - 1; r >= 0; r--) {
        later[r] = nxt;
        if (mask[r]) nxt = (uint32_t)(base + (uint64_t)r * 64 + __builtin_ctzll(mask[r]));
    }
#pragma unroll
    for (int r = 0; r < RLE_ROWS; r++) {
        const uint64_t m = mask[r];
        if ((m >> lane) & 1) {
            const uint32_t pos = (uint32_t)(base + (uint64_t)r * 64 + lane);
            const uint64_t above = (lane == 63) ? 0ull : (m >> (lane + 1));
            const uint32_t next = above ? pos + 1 + (uint32_t)__builtin_ctzll(above) : later[r];
            const uint32_t j = out + __popcll(m & psk_lanemask_lt(lane));
            words[j] = key[r];
            freqs[j] = next - pos;
        }
        out += __popcll(m);
    }
}

// binary search of `n` query words in a sorted list; 0 if absent
__global__ void lookup_counts_kernel(const uint64_t *__restrict__ words, const uint32_t *__restrict__ freqs,
                                     uint64_t nu, const uint64_t *__restrict__ q, uint64_t n,
                                     uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = q[i];
    uint64_t lo = 0, hi = nu;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (words[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[i] = (lo < nu && words[lo] == key) ? freqs[lo] : 0u;
}

// prediction path: every window's canonical word is probed in an open-addressing table of the dictionary; hits bump
// the dictionary entry's counter.  A dictionary of up to DICT_SLOTS / 2 words (the reference's default model has
// 1000 k-mers) keeps the table in LDS; a larger one (`--n_kmers 0` = no limit) is probed in global memory, where it
// stays L2-resident.
constexpr int DICT_SLOTS = 4096;  // power of two, >= 2 * n_dict
__device__ __forceinline__ uint32_t dict_hash(uint64_t w) { return (uint32_t)((w * 0x9E3779B97F4A7C15ull) >> 40); }

template <bool IN_LDS>
__global__ __launch_bounds__(EX_THREADS) void dict_count_kernel(const uint8_t *__restrict__ clean, uint64_t len, int k,
                                                                 const uint64_t *__restrict__ slot_word,
                                                                 const int32_t *__restrict__ slot_idx, uint32_t slot_mask,
                                                                 uint32_t *__restrict__ counts)
{
    __shared__ uint64_t sw[IN_LDS ? DICT_SLOTS : 1];
    __shared__ int32_t si[IN_LDS ? DICT_SLOTS : 1];
    if (IN_LDS) {
        for (int i = threadIdx.x; i < DICT_SLOTS; i += EX_THREADS) { sw[i] = slot_word[i]; si[i] = slot_idx[i]; }
        __syncthreads();
    }
    const uint64_t *tw = IN_LDS ? sw : slot_word;
    const int32_t *ti = IN_LDS ? si : slot_idx;
    const uint64_t g = (uint64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    const uint64_t s = g * EX_SEG;
    if (s >= len) return;
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const int rcshift = 2 * (k - 1);
    Roll r{0, 0, 0};
    const uint64_t start = (s >= (uint64_t)(k - 1)) ? s - (k - 1) : 0;
    for (uint64_t i = start; i < s; i++) roll_byte(r, clean[i], mask, rcshift, k);
    for (int j = 0; j < EX_SEG && s + j < len; j++) {
        roll_byte(r, clean[s + j], mask, rcshift, k);
        if (r.run >= k) {
            const uint64_t w = (r.fw < r.rc) ? r.fw : r.rc;
            uint32_t h = dict_hash(w) & slot_mask;
            while (ti[h] >= 0) {
                if (tw[h] == w) { atomicAdd(&counts[ti[h]], 1u); break; }
                h = (h + 1) & slot_mask;
            }
        }
    }
}

int upload_clean(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint64_t *clean_len);

}  // namespace

int upload_clean_stream(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint64_t *clean_len)
{
    return upload_clean(ctx, bytes, len, clean_len);
}

int launch_extract(psk_ctx *ctx, const uint8_t *clean, uint64_t len, int k, uint64_t lo, uint64_t hi, uint64_t *out,
                   uint32_t *n_out)
{
    if (len == 0) return PSK_OK;
    const uint64_t threads = (len + EX_SEG - 1) / EX_SEG;
    extract_kernel<<<div_up(threads, EX_THREADS), EX_THREADS, 0, ctx->stream>>>(clean, len, k, lo, hi, out, n_out);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

// frames `bytes` into `stage` (host, thread-safe) and pads it for the extract kernel
static int frame_into(uint8_t *stage, size_t stage_cap, const uint8_t *bytes, size_t len, uint64_t *clean_len,
                      uint64_t *padded_len, int k = 0, uint64_t *n_windows = nullptr)
{
    int64_t n = frame_sequence_counting(bytes, len, stage, stage_cap, k, n_windows);
    if (n < 0) return (int)n;
    const uint64_t padded = ((uint64_t)n + EX_SEG - 1) / EX_SEG * EX_SEG + EX_SEG;
    memset(stage + n, '\n', padded - n);
    *clean_len = (uint64_t)n;
    *padded_len = padded;
    return PSK_OK;
}

// ---- pipelined form of the GPU half (psk_count_kmers_batch) ------------------------------------------------------------
// The window count is known from the framing (or bounded by it), so nothing has to come back from the GPU before a chain is
// launched; the only values the host needs -- the number of unique words, for the arena allocation -- are picked up
// late.  Two shapes of the loop in count_batch_impl:
//   one sample per chain (read sets, k >= 17, prediction's consumer, PSK_DC_GROUP=1): chain i is queued on buffer set i % 3,
//     then sample i - 1 is finalised (its event has long fired while chain i keeps the GPU busy); samples i + 1 and i + 2
//     are uploaded and framed ahead on their own streams;
//   groups of G = 8 genomes per chain (k <= 16, r03): 3 G buffer sets, a group's uploads and framing run ahead, one
//     launch chain counts the group (dense_group_enqueue / bucket_group_enqueue), the group before it is finalised meanwhile.
// uploads rotate over two copy streams (PSK_COPY_STREAMS = 1..4): with one, the next copy is only queued when the
// previous one has gone -- 256 genomes took 38-40 ms, with two 33-36 (r03; a stream per buffer set: set index mod streams)
static int copy_stream_count()
{
    static const int n_cs = [] {
        const char *e = getenv("PSK_COPY_STREAMS");
        const int v = e ? atoi(e) : 2;
        return v < 1 ? 1 : (v > 4 ? 4 : v);
    }();
    return n_cs;
}

static int lane_prepare(psk_ctx *ctx, CountLane &L)
{
    if (!L.done) {
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.raw_ready, hipEventDisableTiming));
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.raw_free, hipEventDisableTiming));
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.up_done, hipEventDisableTiming));
    }
    if (!L.pinned_cnt) PSK_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&L.pinned_cnt), 64, hipHostMallocDefault));
    if (!ctx->copy_stream || !ctx->frame_stream) {
        // A process's first upload: the streams of the ingest.  hipStreamCreate costs ~7.5 ms apiece (a hardware queue), and all
        // five of r03 (copy + three more + framing, whatever PSK_COPY_STREAMS said) were created here one after the other: 40 of
        // the 55 ms of a process's first counting call (r04, PSK_TRACE).  Only the copy streams in use now, and together
        std::vector<hipStream_t *> want;
        if (!ctx->copy_stream) want.push_back(&ctx->copy_stream);
        for (int c = 0; c + 1 < copy_stream_count() && c < 3; c++) if (!ctx->copy_more[c]) want.push_back(&ctx->copy_more[c]);
        if (!ctx->frame_stream) want.push_back(&ctx->frame_stream);
        std::vector<hipError_t> err(want.size(), hipSuccess);
        std::vector<std::thread> th;
        for (size_t q = 1; q < want.size(); q++)
            th.emplace_back([&, q] {
                err[q] = hipSetDevice(ctx->device);
                if (err[q] == hipSuccess) err[q] = hipStreamCreateWithFlags(want[q], hipStreamNonBlocking);
            });
        if (!want.empty()) err[0] = hipStreamCreateWithFlags(want[0], hipStreamNonBlocking);
        for (auto &t : th) t.join();
        for (hipError_t e : err)
            if (e != hipSuccess) return psk_fail(ctx, PSK_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    return PSK_OK;
}

int frame_gpu_enqueue(psk_ctx *ctx, hipStream_t stream, int format, const uint8_t *d_raw, uint64_t raw_len, uint8_t *d_clean,
                      void *scratch, uint64_t *host_out);   // frame_gpu.hip
size_t frame_gpu_scratch_bytes(uint64_t raw_len);
int frame_probe(const uint8_t *bytes, size_t len, size_t *start, size_t *end);
int frame_probe_known_end(const uint8_t *bytes, size_t nul_at, size_t *start, size_t *end);

// A grouped batch rotates 3 G buffer sets of eight device buffers each: carved out of ONE allocation (and one pinned block
// for the sets' counters) sized for the batch's longest sample, instead of ~170 hipMallocs on a cold context.  Buffers a
// set already owns and that are large enough stay as they are.  Nothing of an earlier batch is in flight here.
static void lane_set_wants(psk_ctx *ctx, size_t max_len, bool gpu_framing, size_t want[8])
{
    size_t dcb[5];
    if (ctx->dense_mode) dense_lane_bytes(ctx, max_len, dcb);
    else bucket_lane_bytes(ctx, max_len, dcb);
    const size_t w[8] = {max_len + 128 + 2 * EX_SEG, gpu_framing ? max_len + 64 : 0, gpu_framing ? frame_gpu_scratch_bytes(max_len) : 0,
                         dcb[0], dcb[1], dcb[2], dcb[3], dcb[4]};
    for (int q = 0; q < 8; q++) want[q] = w[q];
}
// bytes of ONE buffer set of a grouped batch whose longest sample has max_len bytes
static size_t lane_set_bytes(psk_ctx *ctx, size_t max_len, bool gpu_framing)
{
    size_t want[8], per_lane = 0;
    lane_set_wants(ctx, max_len, gpu_framing, want);
    for (int q = 0; q < 8; q++) per_lane += (want[q] + want[q] / 8 + 511) & ~size_t(255);
    return per_lane;
}
// the slices carved out of the slab are forgotten (before a new layout, and when psk_begin gives a large slab back)
void psk_forget_lane_slices(psk_ctx *ctx)
{
    for (CountLane &L : ctx->lane) {
        DevBuf *b[8] = {&L.raw, &L.rawin, &L.fr_scratch, &L.dc_part, &L.dc_wgoff, &L.dc_cnt, &L.dc_meta, &L.dc_mtemp};
        for (int q = 0; q < 8; q++)
            if (b[q]->borrowed) { b[q]->p = nullptr; b[q]->cap = 0; b[q]->borrowed = false; }
    }
}

static int carve_lanes(psk_ctx *ctx, int n_lanes, size_t max_len, bool gpu_framing)
{
    size_t want[8];
    lane_set_wants(ctx, max_len, gpu_framing, want);
    auto bufs_of = [](CountLane &L, DevBuf *out[8]) {
        out[0] = &L.raw; out[1] = &L.rawin; out[2] = &L.fr_scratch; out[3] = &L.dc_part; out[4] = &L.dc_wgoff; out[5] = &L.dc_cnt;
        out[6] = &L.dc_meta; out[7] = &L.dc_mtemp;
    };
    size_t per_lane = 0;
    for (size_t w : want) per_lane += (w + w / 8 + 511) & ~size_t(255);
    const size_t total = per_lane * (size_t)n_lanes;
    bool need = false;
    for (int l = 0; l < n_lanes && !need; l++) {
        DevBuf *b[8];
        bufs_of(ctx->lane[l], b);
        for (int q = 0; q < 8; q++) need = need || (want[q] && !(b[q]->p && b[q]->cap >= want[q]));
    }
    if (need) {
        // a new layout: every buffer carved out of the slab so far is forgotten first (the slices of two layouts overlap)
        psk_forget_lane_slices(ctx);
        PSK_TRY(dev_reserve(ctx, ctx->lane_slab, total));
        // carve: set l takes slice l; a buffer that is its set's own and large enough is left alone
        for (int l = 0; l < n_lanes; l++) {
            DevBuf *b[8];
            bufs_of(ctx->lane[l], b);
            size_t off = per_lane * (size_t)l;
            for (int q = 0; q < 8; q++) {
                const size_t sz = (want[q] + want[q] / 8 + 511) & ~size_t(255);
                if (want[q] && !(b[q]->p && b[q]->cap >= want[q])) {
                    if (b[q]->p && !b[q]->borrowed) (void)hipFree(b[q]->p);
                    b[q]->p = static_cast<uint8_t *>(ctx->lane_slab.p) + off;
                    b[q]->cap = sz;
                    b[q]->borrowed = true;
                    if (q == 5) ctx->lane[l].dc_slot = 0;   // a fresh counter ring: zeroed before its first use
                }
                off += sz;
            }
        }
    }
    if (!ctx->lane_pinned) PSK_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->lane_pinned), (size_t)16 * 4 * psk_ctx::LANES, hipHostMallocDefault));
    for (int l = 0; l < n_lanes; l++)
        if (!ctx->lane[l].pinned_cnt) ctx->lane[l].pinned_cnt = ctx->lane_pinned + 16 * l;
    return PSK_OK;
}

// {clean length, irregular flag} of a sample framed on the GPU land here (pinned, behind the lane's counters)
static inline uint64_t *lane_frame_result(CountLane &L) { return reinterpret_cast<uint64_t *>(L.pinned_cnt + 8); }

// Stage A of a sample on buffer set L, on the copy stream: the upload and, for raw file bytes (format 1 FASTA,
// 2 FASTQ; frame_gpu.hip), the framing kernels that turn them into the clean stream.  format 0: `src` is a clean
// stream the host framed, `bytes` its padded length.  L.raw_ready fires when the clean stream is in L.raw.
static int chain_upload(psk_ctx *ctx, CountLane &L, const uint8_t *src, uint64_t bytes, int format, bool src_on_device = false)
{
    PSK_TRY(lane_prepare(ctx, L));
    if (bytes == 0) return PSK_OK;
    if (bytes >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample larger than 4 GB");
    const int which = (int)((&L - ctx->lane) % copy_stream_count());
    hipStream_t cs = which ? ctx->copy_more[which - 1] : ctx->copy_stream;
    // after the last reader of this set's clean stream (the sample before last)
    if (L.raw_used) PSK_HIP(ctx, hipStreamWaitEvent(cs, L.raw_free, 0));
    if (format == 0) {
        PSK_TRY(dev_reserve(ctx, L.raw, bytes));
        PSK_HIP(ctx, hipMemcpyAsync(L.raw.p, src, bytes, hipMemcpyHostToDevice, cs));
    } else {
        PSK_TRY(dev_reserve(ctx, L.rawin, bytes + 64));
        PSK_TRY(dev_reserve(ctx, L.raw, bytes + 128));
        PSK_TRY(dev_reserve(ctx, L.fr_scratch, frame_gpu_scratch_bytes(bytes)));
        PSK_HIP(ctx, hipMemcpyAsync(L.rawin.p, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs));   // (a .gz sample: its text was inflated on the device)
        // the framing kernels run on their own stream: the copy stream goes on with the next sample's upload (PCIe is
        // the slowest stage of the ingest: ~100 us per 5-Mbp sample against ~25 us of framing and ~60 us of counting)
        PSK_HIP(ctx, hipEventRecord(L.up_done, cs));
        PSK_HIP(ctx, hipStreamWaitEvent(ctx->frame_stream, L.up_done, 0));
        PSK_TRY(frame_gpu_enqueue(ctx, ctx->frame_stream, format, L.rawin.as<uint8_t>(), bytes, L.raw.as<uint8_t>(), L.fr_scratch.p,
                                  lane_frame_result(L)));
        PSK_HIP(ctx, hipEventRecord(L.raw_ready, ctx->frame_stream));
        return PSK_OK;
    }
    PSK_HIP(ctx, hipEventRecord(L.raw_ready, cs));
    return PSK_OK;
}

// the buffers of the radix route (and of the bucketed sort's fall-back onto it)
static int radix_lane_reserve(psk_ctx *ctx, CountLane &L, uint64_t n)
{
    PSK_TRY(dev_reserve(ctx, L.keysA, n * 8));
    PSK_TRY(dev_reserve(ctx, L.keysB, n * 8));
    PSK_TRY(dev_reserve(ctx, L.starts, (size_t)div_up(n, RLE_TILE) * 8));  // tile offsets | next-head positions
    PSK_TRY(dev_reserve(ctx, L.cnt, (size_t)CountLane::CNT_SLOTS * 16));
    return PSK_OK;
}

// Stage B: the counting chain of the sample whose clean stream stage A put (or is putting) into L.raw.
// n = number of k-base windows (exact from the host's framing; the clean length, an upper bound, after the GPU's).
static int chain_compute(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n, bool n_exact)
{
    ctx->lists[sample_idx] = SampleList();
    ctx->have_presence = false;
    if (clean_len >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample larger than 4 Gbases");
    L.sample = sample_idx;
    L.n = n;
    L.exact = n_exact && ctx->slab_lo == 0 && ctx->slab_hi == 0;  // no slab filter: every window yields a word
    L.uniq = nullptr;
    L.dense = false;
    L.bs = false;
    if (n == 0) {
        if (ctx->dense_mode) {   // an empty sample still owns a (zero) bitmap: the presence build reads every sample's
            SampleList &S = ctx->lists[sample_idx];
            const size_t bytes = (size_t)ctx->dense_nb * DC_BUCKET_WORDS * 8;
            PSK_TRY(arena_alloc(ctx, bytes, (void **)&S.bitmap));
            PSK_HIP(ctx, hipMemsetAsync(S.bitmap, 0, bytes, ctx->stream));
            S.dense = true;
        }
        return PSK_OK;
    }
    PSK_HIP(ctx, hipStreamWaitEvent(ctx->stream, L.raw_ready, 0));
    if (ctx->dense_mode) {   // 2k <= 26: no sort (dense_count.hip)
        if (ctx->dense_defer && dense_group_ok(ctx, n)) {   // a genome of a batch: its chain is launched with its group's
            L.group_pending = true;
            L.clean_len = clean_len;
            return PSK_OK;
        }
        return dense_chain_enqueue(ctx, L, sample_idx, clean_len, n);
    }
    if (bucket_route_ok(ctx, n)) {   // k = 14..16, splitters known
        if (ctx->dense_defer) {      // a genome of a batch: its chain is launched with its group's
            L.group_pending = true;
            L.clean_len = clean_len;
            return PSK_OK;
        }
        return bucket_chain_enqueue(ctx, L, sample_idx, clean_len, n);
    }
    PSK_TRY(radix_lane_reserve(ctx, L, n));
    // every sample takes a fresh pre-zeroed counter slot (a 16-byte memset per sample is a 6 us launch)
    if (L.cnt_slot == 0 || L.cnt_slot >= CountLane::CNT_SLOTS) {
        PSK_HIP(ctx, hipMemsetAsync(L.cnt.p, 0, (size_t)CountLane::CNT_SLOTS * 16, ctx->stream));
        L.cnt_slot = 0;
    }
    uint32_t *d_n = L.cnt.as<uint32_t>() + 4 * (size_t)L.cnt_slot++;
    PSK_TRY(launch_extract(ctx, L.raw.as<uint8_t>(), clean_len, ctx->k, ctx->slab_lo, ctx->slab_hi, L.keysA.as<uint64_t>(),
                           d_n));
    PSK_HIP(ctx, hipEventRecord(L.raw_free, ctx->stream));
    L.raw_used = true;
    uint64_t *sorted = nullptr;
    // with a slab filter only the GPU knows how many words were kept: the launches cover the host's count
    // (every window) and the kernels read the real one from d_n[0]
    const uint32_t *n_dev = L.exact ? nullptr : d_n;
    PSK_TRY(dev_radix_sort_u64(ctx, L.keysA.as<uint64_t>(), L.keysB.as<uint64_t>(), n, 0, 2 * ctx->k, &sorted, n_dev));
    const uint32_t n_tiles = (uint32_t)div_up(n, RLE_TILE);
    uint32_t *t_off = L.starts.as<uint32_t>(), *t_next = t_off + n_tiles;
    rle_tile_kernel<<<n_tiles, RLE_THREADS, 0, ctx->stream>>>(sorted, n, n_dev, t_off, t_next);
    PSK_HIP(ctx, hipGetLastError());
    rle_tile_scan_kernel<<<1, 1024, 0, ctx->stream>>>(t_off, t_next, n_tiles, (uint32_t)n, n_dev, d_n + 1);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(L.pinned_cnt, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipEventRecord(L.done, ctx->stream));
    L.uniq = sorted;  // the emit pass (chain_finalize) reads the sorted keys once the output size is known
    return PSK_OK;
}

// second half: arena allocation + the emit pass (straight into the arena) of the sample whose chain ran on this set
static int chain_finalize(psk_ctx *ctx, CountLane &L)
{
    if (L.sample < 0) return PSK_OK;
    SampleList &S = ctx->lists[L.sample];
    const int sample = L.sample;
    const uint64_t windows = L.n;
    L.sample = -1;
    uint64_t nu = 0, n_kept = 0;
    if (L.n > 0) {
        PSK_HIP(ctx, hipEventSynchronize(L.done));
        const uint64_t n_gpu = L.pinned_cnt[0];
        if (L.exact ? (n_gpu != L.n) : (n_gpu > L.n))
            return psk_fail(ctx, PSK_ESTATE, "sample %d: the GPU kept %llu windows, the framing counted %llu", sample,
                            (unsigned long long)n_gpu, (unsigned long long)L.n);
        n_kept = n_gpu;
        nu = L.pinned_cnt[1];
        if (L.dense) {
            L.sample = sample;
            const int rc = dense_chain_finalize(ctx, L, &n_kept, &nu);
            L.sample = -1;
            if (rc != PSK_OK) return rc;
            S.n_unique = nu;
            S.n_total = n_kept;
            S.done = true;
            return PSK_OK;
        }
        if (L.bs) {
            bool fell_back = false;
            PSK_TRY(bucket_chain_finalize(ctx, L, S, n_kept, nu, &fell_back));
            if (!fell_back) {
                S.n_unique = nu;
                S.n_total = n_kept;
                S.done = true;
                return PSK_OK;
            }
            // a bucket outgrew the LDS sort: the partitioned words through the radix sort and the run-length passes, waited for
            // (rare: a sample unlike the one the splitters were taken from)
            uint64_t *sorted = nullptr;
            L.dc_defer_compact = false;   // (in a group: this sample's list is made here, not by the group's packing launch)
            PSK_TRY(radix_lane_reserve(ctx, L, L.n));
            PSK_TRY(bucket_fallback_keys(ctx, L, n_kept, L.keysA.as<uint64_t>()));
            PSK_TRY(dev_radix_sort_u64(ctx, L.keysA.as<uint64_t>(), L.keysB.as<uint64_t>(), n_kept, 0, 2 * ctx->k, &sorted, nullptr));
            uint32_t *d_n = L.cnt.as<uint32_t>() + 4 * (size_t)(CountLane::CNT_SLOTS - 1);
            const uint32_t nt = (uint32_t)div_up(L.n, RLE_TILE);
            uint32_t *t_off = L.starts.as<uint32_t>(), *t_next = t_off + nt;
            if (n_kept) {
                rle_tile_kernel<<<(uint32_t)div_up(n_kept, RLE_TILE), RLE_THREADS, 0, ctx->stream>>>(sorted, n_kept, nullptr, t_off, t_next);
                rle_tile_scan_kernel<<<1, 1024, 0, ctx->stream>>>(t_off, t_next, (uint32_t)div_up(n_kept, RLE_TILE), (uint32_t)n_kept, nullptr,
                                                                 d_n + 1);
            }
            PSK_HIP(ctx, hipGetLastError());
            PSK_HIP(ctx, hipMemcpyAsync(L.pinned_cnt + 1, d_n + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
            PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            nu = n_kept ? L.pinned_cnt[1] : 0;
            L.uniq = sorted;
        }
        PSK_TRY(arena_alloc(ctx, nu * 8, (void **)&S.words));
        PSK_TRY(arena_alloc(ctx, nu * 4, (void **)&S.freqs));
        const uint32_t n_tiles = (uint32_t)div_up(L.n, RLE_TILE);  // the layout of the tile arrays follows L.n
        const uint32_t *t_off = L.starts.as<uint32_t>();
        if (n_kept)
            rle_emit_kernel<<<(uint32_t)div_up(n_kept, RLE_TILE), RLE_THREADS, 0, ctx->stream>>>(L.uniq, n_kept, t_off, t_off + n_tiles,
                                                                                        S.words, S.freqs);
        PSK_HIP(ctx, hipGetLastError());
    }
    S.n_unique = nu;
    S.n_total = n_kept;
    S.done = true;
    PSK_TRY(bucket_splitters_from(ctx, S, windows));   // k = 14..16: the later samples of the run take the bucketed sort
    return PSK_OK;
}

static int ensure_pinned(psk_ctx *ctx, void **buf, size_t *cap, size_t need)
{
    if (need <= *cap && *buf) return PSK_OK;
    pinned_release(ctx, *buf, *cap);   // (r06: pinned buffers come from, and go back to, a process-wide cache: api.hip)
    *buf = nullptr;
    *cap = 0;
    return pinned_acquire(ctx, need, buf, cap);
}

// `consumer` != nullptr (prediction: count_dict_impl): the framed clean stream of sample i on buffer set L goes to it instead
// of the counting chain -- consumer(L, i, clean_len) queues its kernel on ctx->stream -- and no list is made; k_window is
// then the k of the windows (psk_begin need not have been called)
typedef std::function<int(CountLane &, int, uint64_t)> StreamConsumer;
static int count_batch_impl(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes, const char *const *paths,
                            const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads, int sketch_k,
                            int sketch_size, uint32_t sketch_seed, uint64_t *hashes_out, uint64_t *n_hashes_out,
                            int k_window = 0, const StreamConsumer *consumer = nullptr);

// one sample = a batch of one: the same framing (on the GPU for FASTA / four-line FASTQ) and the same chain
extern "C" int psk_count_kmers(psk_ctx *ctx, int sample_idx, const uint8_t *bytes, size_t len, uint64_t *n_unique,
                               uint64_t *n_total)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->k == 0) return psk_fail(ctx, PSK_ESTATE, "psk_begin has not been called");
    if (sample_idx < 0 || sample_idx >= ctx->n_samples) return psk_fail(ctx, PSK_EINVAL, "sample_idx out of range");
    if (!bytes && len) return psk_fail(ctx, PSK_EINVAL, "null input");
    static const uint8_t none = 0;
    const uint8_t *one = bytes ? bytes : &none;
    return count_batch_impl(ctx, sample_idx, 1, &one, nullptr, &len, n_unique, n_total, 1, 0, 0, 0, nullptr, nullptr);
}

// Batch form: `n_threads` host threads frame samples ahead into a ring of pinned buffers while the calling
// thread drives the GPU half of the samples in order, so host tokenisation overlaps device work.
int sketch_enqueue(psk_ctx *ctx, CountLane &L, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size, uint32_t seed);
int sketch_collect(psk_ctx *ctx, CountLane &L, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size,
                   uint32_t seed, uint64_t *hashes_out, uint64_t *n_out);
int sketch_from_device(psk_ctx *ctx, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size, uint32_t seed,
                       uint64_t *hashes_out, uint64_t *n_out);  // minhash.hip

extern "C" int psk_count_kmers_batch(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes,
                                     const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads)
{
    return psk_count_kmers_batch_sketch(ctx, first_sample_idx, n, bytes, lens, n_unique, n_total, n_threads, 0, 0, 0, nullptr,
                                        nullptr);
}

// Same, and (sketch_k > 0) the Mash-compatible MinHash sketch of every sample from the clean stream that is
// already on the device for counting -- the `-w` path needs both and the host frames each file once.
// reads a whole file into `buf` (worker thread); 0 on success
static int read_whole_file(const char *path, size_t expect, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    buf.resize(expect ? expect : 1);
    size_t got = 0;
    while (got < expect) {
        const size_t r = fread(buf.data() + got, 1, expect - got, f);
        if (r == 0) break;
        got += r;
    }
    fclose(f);
    return got == expect ? 0 : -1;
}

// Large samples (read sets: hundreds of MB) are moved into the pinned slot by several threads -- one thread copies
// at ~10 GB/s, which was the whole cost of a 0.63-GB FASTQ sample (60 ms of 61) -- and the same threads look for the
// NUL that ends the input.  src != nullptr: memcpy; else pread of `path`.  Returns 0, *nul_at = first NUL or len.
static int parallel_fill(uint8_t *dst, const uint8_t *src, const char *path, size_t len, int helpers, size_t *nul_at)
{
    if (helpers < 1) helpers = 1;
    if (helpers > 16) helpers = 16;
    int fd = -1;
    if (!src) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return -1;
    }
    std::vector<size_t> nul((size_t)helpers, len);
    std::vector<int> bad((size_t)helpers, 0);
    const size_t slice = ((len + helpers - 1) / helpers + 4095) & ~size_t(4095);
    auto work = [&](int h) {
        const size_t lo = (size_t)h * slice, hi = lo + slice < len ? lo + slice : len;
        if (lo >= hi) return;
        if (src) memcpy(dst + lo, src + lo, hi - lo);
        else {
            size_t got = lo;
            while (got < hi) {
                const ssize_t r = pread(fd, dst + got, hi - got, (off_t)got);
                if (r <= 0) { bad[h] = 1; return; }
                got += (size_t)r;
            }
        }
        const void *z = memchr(dst + lo, 0, hi - lo);
        if (z) nul[h] = (size_t)(static_cast<const uint8_t *>(z) - dst);
    };
    std::vector<std::thread> ts;
    for (int h = 1; h < helpers; h++) ts.emplace_back(work, h);
    work(0);
    for (auto &t : ts) t.join();
    if (fd >= 0) close(fd);
    size_t first = len;
    for (int h = 0; h < helpers; h++) { if (bad[h]) return -1; if (nul[h] < first) first = nul[h]; }
    *nul_at = first;
    return 0;
}

// `paths` != nullptr: sample i is the file paths[i] of lens[i] bytes, read by the framing thread that takes it (plain
// FASTA / FASTQ; compressed inputs come through the in-memory form after the host has inflated them)
// gzs (may be null): sample i with gzs[i].dev != nullptr is a .gz input whose text is already on the device
struct GzSample {
    const uint8_t *dev = nullptr;
    int fmt = 0;                 // as frame_probe reports
    uint64_t roff = 0, rlen = 0; // its records
};
static int count_batch_core(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes, const char *const *paths,
                            const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads, int sketch_k,
                            int sketch_size, uint32_t sketch_seed, uint64_t *hashes_out, uint64_t *n_hashes_out, int k_window,
                            const StreamConsumer *consumer, const GzSample *gzs)
{
    if (!ctx) return PSK_EINVAL;
    if (sketch_k != 0) {
        if (sketch_k < 1 || sketch_k > 32 || sketch_size < 1) return psk_fail(ctx, PSK_EINVAL, "bad sketch parameters");
        if (!hashes_out || !n_hashes_out) return psk_fail(ctx, PSK_EINVAL, "null sketch buffers");
    }
    if (!consumer) {
        if (ctx->k == 0) return psk_fail(ctx, PSK_ESTATE, "psk_begin has not been called");
        if (n < 0 || first_sample_idx < 0 || first_sample_idx + n > ctx->n_samples)
            return psk_fail(ctx, PSK_EINVAL, "sample range out of bounds");
    }
    if (n == 0) return PSK_OK;
    if ((!bytes && !paths) || !lens) return psk_fail(ctx, PSK_EINVAL, "null input");
    auto on_device = [&](int i) { return gzs && gzs[i].dev != nullptr; };
    auto bytes_of = [&](int i) -> const uint8_t * { return bytes ? bytes[i] : nullptr; };
    auto path_of = [&](int i) -> const char * { return (bytes && bytes[i]) || !paths ? nullptr : paths[i]; };
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    int fill_helpers = n_threads >= 2 * n ? n_threads / n : (n == 1 ? 8 : 1);   // threads per large sample's copy
    // every sample takes the pipelined path (chain_upload / chain_compute); with a slab filter the host's window count is
    // an upper bound and the kernels read the number of kept words from device memory
    size_t max_len = 0;
    for (int i = 0; i < n; i++) {
        if (!on_device(i) && !bytes_of(i) && !path_of(i) && lens[i]) return psk_fail(ctx, PSK_EINVAL, "null input %d", i);
        if (lens[i] > max_len) max_len = lens[i];
    }
    // read sets (hundreds of MB a sample): the copies into pinned memory share the host's memory bandwidth, so six of
    // them at once all finish late and the first upload waits for them (r02: 6 x 0.63 GB, 87 ms before the first byte
    // moved, 137 ms in all).  One sample at a time with every thread on its copy finishes a sample every ~15 ms and the
    // uploads run beside the next copies.
    if (max_len >= (64u << 20) && n > 1) {
        fill_helpers = n_threads > 8 ? 8 : n_threads;
        n_threads = 1;
    }
    if (n_threads > n) n_threads = n;
    // pinned ring: at most ~4 GiB of it (read-scale FASTQ samples are hundreds of MB each)
    while (n_threads > 1 && (size_t)(n_threads + 4) * max_len > (4ull << 30)) n_threads--;
    // genomes at k <= 13 (dense counting) go through the counting kernels in groups of G: one launch chain per group
    const bool bucket_run = !ctx->dense_mode && ctx->k >= 14 && ctx->k <= 32 && !getenv("PSK_NO_BUCKET_SORT");
    int G = (!consumer && (ctx->dense_mode || bucket_run) && n > 1 && max_len < (64u << 20)) ? dense_group_size() : 1;
    if (G > 1) {
        // The 3 G buffer sets of a grouped batch are one slab sized for the batch's longest sample: ~16 bytes per base on the
        // bucketed route -- 1.8 GB for 5-Mbp genomes, but 23 GB for samples just under the 64-MB grouping limit.
        // Smaller groups when the slab would pass a quarter of the device's free memory (what the context already holds
        // counts as free) or 8 GiB; G = 1 is the one-sample chain, which needs no slab at all.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
        size_t cap = (free_b + ctx->lane_slab.cap) / 4;
        if (cap > ((size_t)8 << 30)) cap = (size_t)8 << 30;
        // (frames on the GPU unless PSK_HOST_FRAMING: the larger of the two layouts is budgeted)
        const size_t per_set = lane_set_bytes(ctx, max_len, true);
        while (G > 1 && (size_t)3 * G * per_set > cap) G /= 2;
    }
    const bool grouped = G > 1;
    const int NL = grouped ? 3 * G : 3;   // buffer sets in rotation
    const int want_ring = grouped ? G + n_threads + 4 : n_threads + 4;   // (grouped: a slot is released as soon as its upload is over)
    const int R = n < want_ring ? n : want_ring;  // ring slots: two (groups) being uploaded / framed ahead, one whose chain is in
                                                  // flight, one released late -- never more than there are samples
    if ((int)ctx->ring.size() < R) { ctx->ring.resize(R, nullptr); ctx->ring_cap.resize(R, 0); }
    // A slot that is missing or too small is pinned by the worker that first fills it (sample i < R is the first user of slot
    // i, and nobody else touches the slot before that sample is ready) -- r04: all R slots up front, ~2 ms of page pinning each,
    // were 16-40 ms in front of the first upload of a process's first calls (20 slots for 5-Mbp genomes); now the first upload
    // waits for one slot and the others are pinned beside it, by the threads that would otherwise wait for their turn
    size_t max_host_len = 0;   // (the text of a .gz sample inflated on the device needs no pinned slot)
    for (int i = 0; i < n; i++)
        if (!on_device(i) && lens[i] > max_host_len) max_host_len = lens[i];
    bool any_host = false;
    for (int i = 0; i < n; i++) any_host = any_host || !on_device(i);
    const size_t slot_need = any_host ? max_host_len + 2 * EX_SEG : 0;
    // FASTA and four-line FASTQ are framed on the GPU (frame_gpu.hip): the worker threads then only move file bytes
    // into pinned memory.  PSK_HOST_FRAMING=1 keeps the host state machine for everything (A/B runs, tests).
    const bool gpu_framing = getenv("PSK_HOST_FRAMING") == nullptr;

    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> state(n, 0);          // 0 pending, 1 ready, -1 reading / framing failed
    std::vector<int> fmt(n, 0);            // 0 the ring slot holds a clean stream (host framing); 1 / 2 raw FASTA / FASTQ bytes
    std::vector<char> pre_up(n, 0);        // (grouped batches) a windowless sample's clean stream is already in its set's buffer
    std::vector<uint64_t> clen(n, 0), plen(n, 0), wins(n, 0), roff(n, 0), rlen(n, 0);
    int consumed = 0;                      // samples whose ring slot may be overwritten
    bool abort = false;
    std::atomic<int> next(0);
    const int k = consumer ? k_window : ctx->k;
    // PSK_TRACE: where the calling thread waits (stderr, one line per call)
    const bool trace = getenv("PSK_TRACE") != nullptr;
    double t_worker = 0, t_frame = 0, t_final = 0;
    std::atomic<long long> t_fill_us(0);
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    auto worker = [&]() {
        std::vector<uint8_t> file_buf;  // file image of the sample in hand (paths form, host framing)
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return abort || consumed > i - R; });  // slot i % R is free again
                if (abort) return;
            }
            uint64_t c = 0, p = 0, w = 0, ro = 0, rl = 0;
            int rc = 0, f = 0;
            if (slot_need && i < R && (!ctx->ring[i] || ctx->ring_cap[i] < slot_need)) {
                if (hipSetDevice(ctx->device) != hipSuccess || ensure_pinned(ctx, &ctx->ring[i], &ctx->ring_cap[i], slot_need) != PSK_OK) rc = -3;
            }
            uint8_t *slot = static_cast<uint8_t *>(ctx->ring[i % R]);
            if (rc) {
                // (no pinned memory: reported below)
            } else if (on_device(i)) {
                f = gzs[i].fmt;
                ro = gzs[i].roff;
                rl = gzs[i].rlen;
            } else if (gpu_framing) {
                // file bytes straight into the pinned slot (by several threads when the sample is large and threads
                // are idle); the probe finds where the records start and end
                size_t nul_at = lens[i];
                const auto tf = std::chrono::steady_clock::now();
                rc = parallel_fill(slot, bytes_of(i), path_of(i), lens[i], lens[i] >= (32u << 20) ? fill_helpers : 1, &nul_at);
                t_fill_us += (long long)(std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count() * 1e6);
                if (rc == 0 && path_of(i) && lens[i] >= 2 && slot[0] == 0x1f && slot[1] == 0x8b) rc = -2;   // gzip (PSK_NO_GPU_GZ): the caller inflates
                if (rc == 0) {
                    size_t st = 0, en = 0;
                    f = frame_probe_known_end(slot, nul_at, &st, &en);   // parallel_fill has found the NUL, if any
                    ro = st;
                    rl = en - st;
                }
            } else {
                const uint8_t *src = bytes_of(i);
                if (path_of(i)) {
                    rc = read_whole_file(path_of(i), lens[i], file_buf);
                    src = file_buf.data();
                    if (rc == 0 && lens[i] >= 2 && src[0] == 0x1f && src[1] == 0x8b) rc = -2;
                }
                if (rc == 0) rc = frame_into(slot, ctx->ring_cap[i % R], src, lens[i], &c, &p, k, &w);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                clen[i] = c; plen[i] = p; wins[i] = w; fmt[i] = f; roff[i] = ro; rlen[i] = rl;
                state[i] = rc == -2 ? -2 : rc == -3 ? -3 : rc ? -1 : 1;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; t++) pool.emplace_back(worker);
    int rc = PSK_OK;
    auto release_upto = [&](int upto) {
        {
            std::lock_guard<std::mutex> lk(mu);
            consumed = upto;
            if (rc != PSK_OK) abort = true;
        }
        cv.notify_all();
    };
    auto report = [&](int i) {
        if (n_unique) n_unique[i] = ctx->lists[first_sample_idx + i].n_unique;
        if (n_total) n_total[i] = ctx->lists[first_sample_idx + i].n_total;
    };
    for (CountLane &L : ctx->lane) { L.sample = -1; L.sk_state = 0; }
    auto collect_sketch = [&](int i) -> int {
        CountLane &L = ctx->lane[i % NL];
        if (wins[i] == 0) {
            // no window of the counting k, so nothing was counted -- but the sketch's k may be shorter: take the
            // clean stream (host framing: from the ring slot, still held) through the synchronous route
            n_hashes_out[i] = 0;
            L.sk_state = 0;
            if (clen[i] == 0) return PSK_OK;
            if (fmt[i] == 0 && !pre_up[i]) {
                PSK_TRY(dev_reserve(ctx, L.raw, plen[i]));
                PSK_HIP(ctx, hipMemcpyAsync(L.raw.p, ctx->ring[i % R], plen[i], hipMemcpyHostToDevice, ctx->stream));
            }
            L.sk_state = 2;
        }
        return sketch_collect(ctx, L, L.raw.as<uint8_t>(), clen[i], sketch_k, sketch_size, sketch_seed,
                              hashes_out + (size_t)i * sketch_size, n_hashes_out + i);
    };
    // stage A of sample i (waits for its worker): upload + GPU framing on the copy stream
    auto stage_a = [&](int i) -> int {
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return state[i] != 0; });
            t_worker += since(t0);
            if (state[i] == -2)
                return psk_fail(ctx, PSK_EGZIP, "sample %d (%s) is gzip-compressed: inflate it and use the in-memory call",
                                first_sample_idx + i, path_of(i) ? path_of(i) : "");
            if (state[i] == -3) return psk_fail(ctx, PSK_ENOMEM, "no pinned memory for sample %d (hipHostMalloc of %zu bytes failed)", first_sample_idx + i, slot_need);
            if (state[i] < 0)
                return psk_fail(ctx, PSK_ERANGE, path_of(i) ? "reading or framing sample %d (%s) failed" : "framing of sample %d failed",
                                first_sample_idx + i, path_of(i) ? path_of(i) : "");
        }
        if (on_device(i) && fmt[i]) return chain_upload(ctx, ctx->lane[i % NL], gzs[i].dev + roff[i], rlen[i], fmt[i], true);
        const uint8_t *slot = static_cast<const uint8_t *>(ctx->ring[i % R]);
        if (fmt[i]) return chain_upload(ctx, ctx->lane[i % NL], slot + roff[i], rlen[i], fmt[i]);
        if (wins[i] >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample with more than 2^32 windows");
        return chain_upload(ctx, ctx->lane[i % NL], slot, wins[i] ? plen[i] : 0, 0);
    };
    // stage B: the counting chain, once the length of the clean stream is known on the host
    auto consume = [&](CountLane &L, int i, uint64_t clean_len, uint64_t n_windows, bool exact) -> int {
        if (consumer) return (clean_len && n_windows) ? (*consumer)(L, first_sample_idx + i, clean_len) : PSK_OK;   // (its callers count from 0; a call cut into runs goes on counting)
        return chain_compute(ctx, L, first_sample_idx + i, clean_len, n_windows, exact);
    };
    auto stage_b = [&](int i) -> int {
        CountLane &L = ctx->lane[i % NL];
        if (fmt[i] && rlen[i]) {
            const auto t0 = std::chrono::steady_clock::now();
            PSK_HIP(ctx, hipEventSynchronize(L.raw_ready));   // the GPU is busy with the chain of sample i - 1 meanwhile
            t_frame += since(t0);
            const uint64_t *res = lane_frame_result(L);
            if (fmt[i] == 2 && res[1] && getenv("PSK_HOST_WRAPPED_FASTQ")) {
                // (r05's route, kept as the A/B knob: the host state machine frames a FASTQ sample that is not four-line FASTQ)
                PSK_TRY(ensure_pinned(ctx, &ctx->pinned, &ctx->pinned_cap, rlen[i] + 2 * EX_SEG));
                uint64_t c = 0, p = 0, w = 0;
                std::vector<uint8_t> text;   // (a .gz sample inflated on the device: its text comes back for the host's state machine)
                const uint8_t *records = on_device(i) ? nullptr : static_cast<const uint8_t *>(ctx->ring[i % R]) + roff[i];
                if (!records) {
                    text.resize(rlen[i]);
                    PSK_HIP(ctx, hipMemcpy(text.data(), gzs[i].dev + roff[i], rlen[i], hipMemcpyDeviceToHost));
                    records = text.data();
                }
                const int frc = frame_into(static_cast<uint8_t *>(ctx->pinned), ctx->pinned_cap, records, rlen[i], &c, &p, k, &w);
                if (frc) return psk_fail(ctx, frc, "framing of sample %d failed", first_sample_idx + i);
                clen[i] = c; plen[i] = p; wins[i] = w; fmt[i] = 0;
                PSK_TRY(chain_upload(ctx, L, static_cast<const uint8_t *>(ctx->pinned), w ? p : 0, 0));
                PSK_HIP(ctx, hipEventSynchronize(L.raw_ready));   // ctx->pinned is reused by the next such sample
                return consume(L, i, c, w, true);
            }
            if (fmt[i] == 2 && res[1]) {
                // not four-line FASTQ (records over several lines, blank lines between them): framed again, on the device, by the
                // scan of line kinds (frame_gpu.hip, format 3; r06 -- until then the host's state machine took such a sample); the
                // raw bytes are still in the lane's input buffer
                if (getenv("PSK_TRACE"))
                    fprintf(stderr, "[psk] sample %d: FASTQ, but not four lines a record: framed on the device by the scan of line kinds\n", first_sample_idx + i);
                PSK_TRY(frame_gpu_enqueue(ctx, ctx->frame_stream, 3, L.rawin.as<uint8_t>(), rlen[i], L.raw.as<uint8_t>(), L.fr_scratch.p, lane_frame_result(L)));
                PSK_HIP(ctx, hipEventRecord(L.raw_ready, ctx->frame_stream));
                PSK_HIP(ctx, hipEventSynchronize(L.raw_ready));
                res = lane_frame_result(L);
            }
            clen[i] = res[0];
            wins[i] = res[0];   // an upper bound of the window count: sizes the buffers, the GPU counts the windows
            return consume(L, i, clen[i], wins[i], false);
        }
        return consume(L, i, clen[i], wins[i], true);
    };
    if (grouped) rc = carve_lanes(ctx, NL, max_len, gpu_framing);
    const double t_setup = since(t_call);   // (PSK_TRACE: buffer sets carved, threads started)
    if (rc != PSK_OK) {
    } else if (grouped) {
        // Groups of G samples: a group's uploads and framing run two groups ahead on the copy / framing streams; its samples'
        // host halves (the framed length, the arena blocks) are done one by one, then ONE launch chain counts the group
        // (dense_group_enqueue); the group before it is finalised meanwhile -- sizes read back, multi-count blocks
        // allocated, one compaction launch, sketches collected -- which frees its buffer sets for the group after next.
        ctx->dense_defer = true;
        int next_a = 0, released = 0;
        auto pump_a = [&](int upto) { while (rc == PSK_OK && next_a < n && next_a < upto) rc = stage_a(next_a++); };
        auto finalize_group = [&](int lo, int hi) -> int {
            CountLane *gl[8];
            int gs[8], cnt = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = lo; i < hi; i++) {
                CountLane &L = ctx->lane[i % NL];
                PSK_TRY(chain_finalize(ctx, L));
                if (L.dc_defer_compact) { gl[cnt] = &L; gs[cnt] = first_sample_idx + i; cnt++; }   // sized and allocated; packed below
            }
            if (cnt) PSK_TRY(ctx->dense_mode ? dense_group_compact(ctx, gl, gs, cnt) : bucket_group_compact(ctx, gl, gs, cnt));
            t_final += since(t0);
            for (int i = lo; i < hi; i++) {
                report(i);
                if (sketch_k) PSK_TRY(collect_sketch(i));
            }
            if (hi > released) { released = hi; release_upto(hi); }
            return PSK_OK;
        };
        pump_a(G);
        int prev_lo = -1, prev_hi = -1;
        for (int lo = 0, hi = 0; lo < n && rc == PSK_OK; lo = hi) {
            // (k = 14..16: the first sample of a run goes through the radix route alone and is finalised at once -- its list
            // gives the splitters of the bucketed sort that every later sample, grouped, takes)
            const bool alone = bucket_run && !ctx->bs_ready;
            hi = alone ? lo + 1 : (lo + G < n ? lo + G : n);
            CountLane *gl[8];
            int gs[8], cnt = 0;
            uint64_t gc[8], gn[8];
            for (int i = lo; i < hi && rc == PSK_OK; i++) {
                rc = stage_b(i);   // a genome's chain stays pending; anything else (an empty sample, a read set) is queued here
                CountLane &L = ctx->lane[i % NL];
                // the pinned slot goes back as soon as the upload is over (framed on the GPU: stage B has waited for the
                // framing; framed on the host: wait for the copy here -- it ran G samples ahead), so the ring needs G + a few
                // slots, not three groups' worth (a cold context pays ~2 ms per pinned slot).  A sample too short for a
                // window of the counting k whose sketch will still want its clean stream: uploaded now.
                if (rc == PSK_OK && !fmt[i] && wins[i] > 0 && hipEventSynchronize(L.raw_ready) != hipSuccess)
                    rc = psk_fail(ctx, PSK_EHIP, "event wait failed");
                if (rc == PSK_OK && sketch_k && wins[i] == 0 && fmt[i] == 0 && clen[i]) {
                    rc = dev_reserve(ctx, L.raw, plen[i]);
                    if (rc == PSK_OK && (hipMemcpyAsync(L.raw.p, ctx->ring[i % R], plen[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                                         hipStreamSynchronize(ctx->stream) != hipSuccess))
                        rc = psk_fail(ctx, PSK_EHIP, "upload of a short sample failed");
                    pre_up[i] = 1;
                }
                if (rc == PSK_OK) { released = i + 1; release_upto(released); }
                // the copy stream is kept G samples ahead (their buffer sets are those of the group before last: finalised)
                pump_a(i + 1 + G);
                if (rc == PSK_OK && L.group_pending) {
                    L.group_pending = false;
                    gl[cnt] = &L; gs[cnt] = first_sample_idx + i; gc[cnt] = L.clean_len; gn[cnt] = L.n;
                    cnt++;
                }
            }
            if (rc == PSK_OK && cnt) rc = ctx->dense_mode ? dense_group_enqueue(ctx, gl, gs, gc, gn, cnt) : bucket_group_enqueue(ctx, gl, gs, gc, gn, cnt);
            for (int i = lo; i < hi && rc == PSK_OK; i++)
                if (sketch_k && wins[i] > 0)
                    rc = sketch_enqueue(ctx, ctx->lane[i % NL], ctx->lane[i % NL].raw.as<uint8_t>(), clen[i], sketch_k, sketch_size, sketch_seed);
            if (rc == PSK_OK && prev_lo >= 0) rc = finalize_group(prev_lo, prev_hi);
            prev_lo = lo; prev_hi = hi;
            if (rc == PSK_OK && alone) { rc = finalize_group(lo, hi); prev_lo = -1; }
        }
        if (rc == PSK_OK && prev_lo >= 0) rc = finalize_group(prev_lo, prev_hi);
        ctx->dense_defer = false;
        for (CountLane &L : ctx->lane) { L.group_pending = false; L.dc_defer_compact = false; }
    } else {
    // Samples i + 1 and i + 2 are uploaded and framed on the copy stream while chain i runs: the host's wait for the
    // framed length of sample i (stage B) then finds it long done (with one sample ahead the wait sat on the critical
    // path: 200 us per 5-Mbp sample instead of 145).
    rc = stage_a(0);
    if (rc == PSK_OK && n > 1) rc = stage_a(1);
    for (int i = 0; i < n && rc == PSK_OK; i++) {
        rc = stage_b(i);
        if (rc == PSK_OK && i > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            if (consumer) {   // no list to finalise: the upload of sample i - 1 has to be over before its ring slot is reused
                if (hipEventSynchronize(ctx->lane[(i - 1) % 3].raw_ready) != hipSuccess) rc = psk_fail(ctx, PSK_EHIP, "event wait failed");
            } else {
                rc = chain_finalize(ctx, ctx->lane[(i - 1) % 3]);  // waits for chain i - 1: its upload is done too
            }
            t_final += since(t0);
            if (rc == PSK_OK && !consumer) report(i - 1);
            if (rc == PSK_OK && sketch_k) rc = collect_sketch(i - 1);
            release_upto(i);
        }
        // the sketch of sample i: queued behind its chain, collected one sample later (after chain i + 1 has been
        // queued, so the stream never runs dry; before sample i + 2 is uploaded into this lane's clean-stream buffer)
        if (rc == PSK_OK && sketch_k && wins[i] > 0)
            rc = sketch_enqueue(ctx, ctx->lane[i % 3], ctx->lane[i % 3].raw.as<uint8_t>(), clen[i], sketch_k, sketch_size, sketch_seed);
        // set (i + 2) % 3 is free again: chain i - 1 has been finalised and its sketch collected
        if (rc == PSK_OK && i + 2 < n) rc = stage_a(i + 2);
    }
    if (rc == PSK_OK && n > 0 && !consumer) {
        rc = chain_finalize(ctx, ctx->lane[(n - 1) % 3]);
        if (rc == PSK_OK) report(n - 1);
        if (rc == PSK_OK && sketch_k) rc = collect_sketch(n - 1);
    }
    }
    {   // nothing of this call may still be in flight when it returns (the ring and the caller's buffers)
        const hipError_t e1 = hipStreamSynchronize(ctx->copy_stream ? ctx->copy_stream : ctx->stream);
        for (hipStream_t cs2 : ctx->copy_more) if (cs2) (void)hipStreamSynchronize(cs2);
        if (ctx->frame_stream) (void)hipStreamSynchronize(ctx->frame_stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (ctx->sketch_stream) (void)hipStreamSynchronize(ctx->sketch_stream);
        if (rc == PSK_OK && (e1 != hipSuccess || e2 != hipSuccess))
            rc = psk_fail(ctx, PSK_EHIP, "stream synchronisation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        for (CountLane &L : ctx->lane) L.sample = -1;
    }
    release_upto(n);
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != PSK_OK) abort = true;
    }
    cv.notify_all();
    for (auto &t : pool) t.join();
    if (trace)
        fprintf(stderr, "[psk] count batch: %d samples %.1f ms; the caller waited %.1f ms for the host threads (their fills: %.1f ms "
                        "in all), %.1f ms for uploads + framing, %.1f ms for chains; set-up %.1f ms\n", n, since(t_call) * 1e3, t_worker * 1e3,
                t_fill_us.load() / 1e3, t_frame * 1e3, t_final * 1e3, t_setup * 1e3);
    return rc;
}

// The batch as the entry points hand it over.  Samples that are gzip images (magic bytes; glistmaker reads .gz through zlib:
// SURVEY.md section 2 row 9) are inflated on the device first (gz_inflate.hip) -- every .gz sample of a run in one go, runs
// cut where the text would pass PSK_GZ_GROUP_MB (8 GiB -- r06; ~1.5 GB of compressed input, 65,536 decoding lanes of 23 KB each: with r05's 12 GiB the inflate's buffers were 70 GB, and what a hipMalloc beyond the first ~40 GB of a process costs on this pool -- 20-30 ms per GB, tools/free_probe.py -- made 64 read sets take 2.1 s to the .pkl where 8-GiB runs take 1.27, 6-GiB 1.4, 4-GiB 1.56: profiles/r06_cfg5gz_groups.json) -- and their chains then start from text that is already in device
// memory; a member the device route declines has been inflated by zlib on the host and goes on as an in-memory sample.
static int count_batch_impl(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes, const char *const *paths,
                            const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads, int sketch_k,
                            int sketch_size, uint32_t sketch_seed, uint64_t *hashes_out, uint64_t *n_hashes_out, int k_window,
                            const StreamConsumer *consumer)
{
    auto core = [&](int lo, int cnt, const uint8_t *const *b, const char *const *p, const size_t *l, const GzSample *g) {
        return count_batch_core(ctx, first_sample_idx + lo, cnt, b, p, l, n_unique ? n_unique + lo : nullptr, n_total ? n_total + lo : nullptr,
                                n_threads, sketch_k, sketch_size, sketch_seed, hashes_out ? hashes_out + (size_t)lo * sketch_size : nullptr,
                                n_hashes_out ? n_hashes_out + lo : nullptr, k_window, consumer, g);
    };
    if (!ctx || n <= 0 || (!bytes && !paths) || !lens || getenv("PSK_NO_GPU_GZ")) return core(0, n, bytes, paths, lens, nullptr);
    std::vector<char> is_gz((size_t)n, 0);
    bool any = false;
    for (int i = 0; i < n; i++) {
        uint8_t m[2] = {0, 0};
        if (lens[i] < 18) continue;
        if (bytes && bytes[i]) {
            m[0] = bytes[i][0];
            m[1] = bytes[i][1];
        } else if (paths && paths[i]) {
            FILE *f = fopen(paths[i], "rb");
            if (f) {
                if (fread(m, 1, 2, f) != 2) m[0] = 0;
                fclose(f);
            }
        }
        is_gz[(size_t)i] = m[0] == 0x1f && m[1] == 0x8b;
        any = any || is_gz[(size_t)i];
    }
    if (!any) return core(0, n, bytes, paths, lens, nullptr);
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->gz_stream) PSK_HIP(ctx, hipStreamCreateWithFlags(&ctx->gz_stream, hipStreamNonBlocking));
    if (!ctx->gz_up_stream) PSK_HIP(ctx, hipStreamCreateWithFlags(&ctx->gz_up_stream, hipStreamNonBlocking));
    const char *gm = getenv("PSK_GZ_GROUP_MB");
    const size_t budget = (size_t)(gm && *gm ? strtoull(gm, nullptr, 10) : 8192) << 20;
    const bool host_only = getenv("PSK_HOST_FRAMING") != nullptr;   // (the A/B knob of the host's state machine: the host's inflate with it)
    const bool trace = getenv("PSK_TRACE") != nullptr;
    std::vector<uint8_t *> held((size_t)n, nullptr);   // where the compressed image of a .gz FILE is (a slice of ctx->gz_host[set])
    std::vector<const uint8_t *> eb((size_t)n, nullptr);
    std::vector<const char *> ep((size_t)n, nullptr);
    std::vector<size_t> el((size_t)n, 0);
    std::vector<GzSample> gs((size_t)n);
    auto image = [&](int i) -> const uint8_t * { return bytes && bytes[i] ? bytes[i] : held[(size_t)i]; };
    // ISIZE of the last member (the text of a one-member file, modulo 2^32): what a run's budget is counted in
    auto isize_of = [&](int i) -> size_t {
        uint8_t d[4] = {0, 0, 0, 0};
        if (bytes && bytes[i]) {
            memcpy(d, bytes[i] + lens[i] - 4, 4);
        } else {
            FILE *f = fopen(paths[i], "rb");
            if (f) {
                if (fseek(f, -4, SEEK_END) != 0 || fread(d, 1, 4, f) != 4) memset(d, 0, 4);
                fclose(f);
            }
        }
        return (size_t)d[0] | ((size_t)d[1] << 8) | ((size_t)d[2] << 16) | ((size_t)d[3] << 24);
    };
    // A run: samples [lo, hi) of the call, cut where the text of its .gz samples would pass the budget.  Its images are read (by the
    // threads the framing would use) and inflated into buffer set `set`; its chains then start from text in device memory.
    struct Run {
        int lo = 0, hi = 0, set = 0;
        std::vector<int> idx;            // its .gz samples
        std::vector<size_t> sizes;       // ... their images' sizes, where they lie in the device buffer
        std::vector<uint64_t> at;
        bool on_device = false;          // the device inflates them (else: zlib on host threads)
        std::vector<GzInflated> res;
        std::chrono::steady_clock::time_point t0;
        double ms_read = 0;
    };
    auto plan = [&](int lo, int set) {
        Run r;
        r.lo = r.hi = lo;
        r.set = set;
        size_t est = 0;
        while (r.hi < n) {
            size_t e = 0;
            if (is_gz[(size_t)r.hi]) {
                e = isize_of(r.hi);
                if (e < 3 * lens[r.hi]) e = 3 * lens[r.hi];
            }
            if (r.hi > lo && est + e > budget) break;
            est += e;
            if (is_gz[(size_t)r.hi]) r.idx.push_back(r.hi);
            r.hi++;
        }
        return r;
    };
    // (up != nullptr: the device buffer of the run's images; file j of the run goes to up + at[j] as soon as it has been read)
    auto read_images = [&](Run &r, uint8_t *up, const uint64_t *at) -> int {
        // r06: a .gz FILE is not read into host memory of the library's own any more -- it is MAPPED (read-only, private), the upload
        // copies out of the mapping (the page cache's pages: no fresh anonymous pages to fault in, 0.3 s per 2 GB), and what the host
        // itself reads of an image -- member headers, trailers, the whole of a file the device declines -- it reads there too.  Giving 5.2 GB
        // of such buffers back cost psk_build_presence 0.5 s of cfg5gz's 2.3 s (on a helper thread the same half second was spent by
        // whoever next touched the address space); an unmapped file gives no page back.  PSK_GZ_READ=1: r05's buffers (and what a
        // file that cannot be mapped gets).
        const bool map_files = !getenv("PSK_GZ_READ");
        std::vector<std::pair<void *, size_t>> &maps = ctx->gz_maps[r.set];
        for (auto &m : maps) munmap(m.first, m.second);   // (the run before last of this set: inflated and counted)
        maps.clear();
        size_t need = 0;
        std::vector<char> mapped((size_t)n, 0);
        for (int i : r.idx) {
            if (bytes && bytes[i]) continue;
            if (map_files && lens[i]) {
                const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
                struct stat sb;
                void *m = MAP_FAILED;
                if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= lens[i]) m = mmap(nullptr, lens[i], PROT_READ, MAP_PRIVATE, fd, 0);
                if (fd >= 0) close(fd);
                if (m != MAP_FAILED) {
                    (void)madvise(m, lens[i], MADV_SEQUENTIAL);
                    maps.push_back({m, lens[i]});
                    held[(size_t)i] = static_cast<uint8_t *>(m);
                    mapped[(size_t)i] = 1;
                    continue;
                }
            }
            need += (lens[i] + 63) & ~(size_t)63;
        }
        uint8_t *&host = ctx->gz_host[r.set];
        size_t &cap = ctx->gz_host_cap[r.set];
        if (need > cap) {
            if (ctx->gz_reaper.joinable()) ctx->gz_reaper.join();
            std::free(host);
            cap = 0;
            host = static_cast<uint8_t *>(std::malloc(need + need / 8));
            if (!host) return psk_fail(ctx, PSK_ENOMEM, "no host memory for %zu bytes of compressed input", need);
            cap = need + need / 8;
        }
        size_t used = 0;
        for (int i : r.idx)
            if (!(bytes && bytes[i]) && !mapped[(size_t)i]) {
                held[(size_t)i] = host + used;
                used += (lens[i] + 63) & ~(size_t)63;
            }
        std::atomic<int> next(0), failed(-1), up_failed(0);
        auto reader = [&]() {
            if (up && hipSetDevice(ctx->device) != hipSuccess) up_failed = 1;
            for (;;) {
                const int j = next.fetch_add(1);
                if (j >= (int)r.idx.size()) return;
                const int i = r.idx[(size_t)j];
                if (bytes && bytes[i]) {
                    if (up && lens[i] && hipMemcpyAsync(up + at[j], bytes[i], lens[i], hipMemcpyHostToDevice, ctx->gz_up_stream) != hipSuccess) up_failed = 1;
                    continue;
                }
                if (mapped[(size_t)i]) {
                    if (up && lens[i] && hipMemcpyAsync(up + at[j], held[(size_t)i], lens[i], hipMemcpyHostToDevice, ctx->gz_up_stream) != hipSuccess) up_failed = 1;
                    continue;
                }
                FILE *f = fopen(paths[i], "rb");
                size_t got = 0;
                if (f) {
                    while (got < lens[i]) {
                        const size_t rd = fread(held[(size_t)i] + got, 1, lens[i] - got, f);
                        if (rd == 0) break;
                        got += rd;
                    }
                    fclose(f);
                }
                if (!f || got != lens[i]) failed = i;
                else if (up && lens[i] && hipMemcpyAsync(up + at[j], held[(size_t)i], lens[i], hipMemcpyHostToDevice, ctx->gz_up_stream) != hipSuccess) up_failed = 1;
            }
        };
        std::vector<std::thread> pool;
        const int nt = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
        for (int t = 1; t < nt && t < (int)r.idx.size(); t++) pool.emplace_back(reader);
        reader();
        for (auto &t : pool) t.join();
        if (failed >= 0) return psk_fail(ctx, PSK_ERANGE, "reading sample %d (%s) failed", first_sample_idx + failed.load(), paths[failed.load()]);
        if (up_failed) return psk_fail(ctx, PSK_EHIP, "uploading the compressed images failed");
        return PSK_OK;
    };
    // stage 1 of a run: its images read (and, for a run the device inflates, uploaded as they arrive, on a stream of their own)
    auto stage_read = [&](Run &r) -> int {
        for (int i = r.lo; i < r.hi; i++) {
            eb[(size_t)i] = bytes ? bytes[i] : nullptr;
            ep[(size_t)i] = paths ? paths[i] : nullptr;
            el[(size_t)i] = lens[i];
            gs[(size_t)i] = GzSample();
        }
        if (r.idx.empty()) return PSK_OK;
        PSK_HIP(ctx, hipSetDevice(ctx->device));
        r.t0 = std::chrono::steady_clock::now();
        for (int i : r.idx) r.sizes.push_back(lens[i]);
        r.on_device = gz_group_on_device((int)r.idx.size(), r.sizes.data(), host_only, n_threads);
        r.at.resize(r.idx.size());
        if (r.on_device) {
            const uint64_t total = gz_image_layout((int)r.idx.size(), r.sizes.data(), r.at.data());
            PSK_TRY(dev_reserve(ctx, ctx->gz_comp[r.set], total));
            PSK_HIP(ctx, hipMemsetAsync(ctx->gz_comp[r.set].p, 0, total, ctx->gz_up_stream));
        }
        PSK_TRY(read_images(r, r.on_device ? ctx->gz_comp[r.set].as<uint8_t>() : nullptr, r.at.data()));
        if (r.on_device) PSK_HIP(ctx, hipStreamSynchronize(ctx->gz_up_stream));
        r.ms_read = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t0).count();
        return PSK_OK;
    };
    // stage 2: inflated, and where each sample's records are
    auto stage_inflate = [&](Run &r) -> int {
        if (r.idx.empty()) return PSK_OK;
        PSK_HIP(ctx, hipSetDevice(ctx->device));
        const auto t1 = std::chrono::steady_clock::now();
        std::vector<const uint8_t *> ptrs;
        for (int i : r.idx) ptrs.push_back(image(i));
        DevBuf &out = ctx->gz_out[r.set];
        PSK_TRY(gz_inflate_group(ctx, (int)r.idx.size(), ptrs.data(), r.sizes.data(), ctx->gz_comp[r.set], ctx->gz_sym, ctx->gz_rec, out, ctx->gz_tab, r.res, nullptr,
                                 host_only, n_threads, ctx->gz_stream, r.on_device));
        std::vector<uint8_t> head;
        for (size_t j = 0; j < r.idx.size(); j++) {
            const int i = r.idx[j];
            GzInflated &g = r.res[j];
            ep[(size_t)i] = nullptr;
            if (g.on_device) {
                // where the records start decides the format (frame_probe): the first bytes of the text come back for that
                const uint8_t *text = out.as<uint8_t>() + g.off;
                const size_t end = g.first_nul < g.len ? g.first_nul : g.len, look = end < 65536 ? end : 65536;
                head.resize(look + 1);
                if (look) {
                    PSK_HIP(ctx, hipMemcpyAsync(head.data(), text, look, hipMemcpyDeviceToHost, ctx->gz_stream));
                    PSK_HIP(ctx, hipStreamSynchronize(ctx->gz_stream));
                }
                size_t st = 0, en = 0;
                const int f = frame_probe_known_end(head.data(), look, &st, &en);
                if (f || look == end) {
                    gs[(size_t)i].dev = text;
                    gs[(size_t)i].fmt = f;
                    gs[(size_t)i].roff = f ? st : end;
                    gs[(size_t)i].rlen = f ? end - st : 0;
                    eb[(size_t)i] = nullptr;
                    el[(size_t)i] = g.len;
                    continue;
                }
                // no record in the first 64 KB: the whole text comes back and takes the in-memory route
                g.host.resize(g.len);
                PSK_HIP(ctx, hipMemcpyAsync(g.host.data(), text, g.len, hipMemcpyDeviceToHost, ctx->gz_stream));
                PSK_HIP(ctx, hipStreamSynchronize(ctx->gz_stream));
                g.on_device = false;
            }
            eb[(size_t)i] = g.host.data();
            el[(size_t)i] = g.host.size();
        }
        if (trace) {
            size_t text = 0, comp = 0;
            for (size_t j = 0; j < r.idx.size(); j++) {
                text += r.res[j].len;
                comp += r.sizes[j];
            }
            fprintf(stderr, "[psk] count batch: samples %d..%d: %zu .gz ones, %.1f MB read in %.1f ms, -> %.1f MB of text in %.1f ms\n", r.lo, r.hi - 1,
                    r.idx.size(), comp / 1e6, r.ms_read, text / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        }
        return PSK_OK;
    };
    auto stage_count = [&](Run &r, std::chrono::steady_clock::time_point t_all) -> int {
        const auto t_core = std::chrono::steady_clock::now();
        const int rc = core(r.lo, r.hi - r.lo, eb.data() + r.lo, ep.data() + r.lo, el.data() + r.lo, gs.data() + r.lo);
        if (trace)
            fprintf(stderr, "[psk] count batch: samples %d..%d counted in %.1f ms; %.1f ms since the call began\n", r.lo, r.hi - 1,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_core).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
        return rc;
    };
    // The runs of the call, and the three stages as a pipeline over them: run k + 2 is read while run k + 1 is inflated while run
    // k is counted -- two buffer sets (run k's are run k - 2's: its images are read when run k - 2 has been inflated, its text is
    // written when run k - 2 has been counted), a thread for each of the first two stages, the calling thread counts.
    std::vector<Run> runs;
    for (int lo = 0; lo < n;) {
        runs.push_back(plan(lo, (int)runs.size() & 1));
        lo = runs.back().hi;
    }
    const int R = (int)runs.size();
    const auto t_all = std::chrono::steady_clock::now();
    if (R == 1 || getenv("PSK_GZ_NO_LOOKAHEAD")) {
        for (Run &r : runs) {
            PSK_TRY(stage_read(r));
            PSK_TRY(stage_inflate(r));
            PSK_TRY(stage_count(r, t_all));
        }
        return PSK_OK;
    }
    std::mutex pm;
    std::condition_variable pcv;
    int read_done = 0, inflate_done = 0, count_done = 0, failed_rc = PSK_OK;
    std::string failed_why;
    auto fail = [&](int rc) {   // (the first failure is the call's; the others stop at their next wait)
        std::lock_guard<std::mutex> lk(pm);
        if (failed_rc == PSK_OK) {
            failed_rc = rc;
            failed_why = psk_error_text(ctx);
        }
        pcv.notify_all();
    };
    auto wait_for = [&](const int &counter, int at_least) {
        std::unique_lock<std::mutex> lk(pm);
        pcv.wait(lk, [&] { return failed_rc != PSK_OK || counter >= at_least; });
        return failed_rc == PSK_OK;
    };
    auto advance = [&](int &counter) {
        std::lock_guard<std::mutex> lk(pm);
        counter++;
        pcv.notify_all();
    };
    std::thread reader([&] {
        for (int k = 0; k < R; k++) {
            if (!wait_for(inflate_done, k - 1)) return;   // (the images of run k - 2 have been used)
            const int rc = stage_read(runs[(size_t)k]);
            if (rc != PSK_OK) return fail(rc);
            advance(read_done);
        }
    });
    std::thread inflater([&] {
        for (int k = 0; k < R; k++) {
            if (!wait_for(read_done, k + 1) || !wait_for(count_done, k - 1)) return;   // (the text of run k - 2 has been counted)
            const int rc = stage_inflate(runs[(size_t)k]);
            if (rc != PSK_OK) return fail(rc);
            advance(inflate_done);
        }
    });
    for (int k = 0; k < R; k++) {
        if (!wait_for(inflate_done, k + 1)) break;
        const int rc = stage_count(runs[(size_t)k], t_all);
        if (rc != PSK_OK) {
            fail(rc);
            break;
        }
        advance(count_done);
    }
    reader.join();
    inflater.join();
    if (failed_rc != PSK_OK) {
        psk_set_error_text(ctx, failed_why);
        return failed_rc;
    }
    return PSK_OK;
}

extern "C" int psk_count_kmers_batch_sketch(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes,
                                            const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads,
                                            int sketch_k, int sketch_size, uint32_t sketch_seed, uint64_t *hashes_out,
                                            uint64_t *n_hashes_out)
{
    return count_batch_impl(ctx, first_sample_idx, n, bytes, nullptr, lens, n_unique, n_total, n_threads, sketch_k, sketch_size,
                            sketch_seed, hashes_out, n_hashes_out);
}

// The same for uncompressed files on disk: the framing threads read them, so no file image crosses the caller's
// language boundary (in Python: no bytes object per sample, no GIL hand-offs).
extern "C" int psk_count_kmers_files(psk_ctx *ctx, int first_sample_idx, int n, const char *const *paths, const size_t *sizes,
                                     uint64_t *n_unique, uint64_t *n_total, int n_threads, int sketch_k, int sketch_size,
                                     uint32_t sketch_seed, uint64_t *hashes_out, uint64_t *n_hashes_out)
{
    return count_batch_impl(ctx, first_sample_idx, n, nullptr, paths, sizes, n_unique, n_total, n_threads, sketch_k, sketch_size,
                            sketch_seed, hashes_out, n_hashes_out);
}

namespace {
// frame into the context's single pinned buffer and upload (dictionary counting, MinHash)
int upload_clean(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint64_t *clean_len)
{
    PSK_TRY(ensure_pinned(ctx, &ctx->pinned, &ctx->pinned_cap, len + 2 * EX_SEG));
    uint64_t padded = 0;
    int rc = frame_into(static_cast<uint8_t *>(ctx->pinned), ctx->pinned_cap, bytes, len, clean_len, &padded);
    if (rc) return psk_fail(ctx, rc, "framing failed");
    PSK_TRY(dev_reserve(ctx, ctx->raw, padded));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->raw.p, ctx->pinned, padded, hipMemcpyHostToDevice, ctx->stream));
    return PSK_OK;
}
}  // namespace

extern "C" int psk_get_list(psk_ctx *ctx, int sample_idx, uint64_t *words, uint32_t *freqs, uint64_t cap)
{
    if (!ctx) return PSK_EINVAL;
    if (sample_idx < 0 || sample_idx >= ctx->n_samples || !ctx->lists[sample_idx].done)
        return psk_fail(ctx, PSK_ESTATE, "sample %d has not been counted", sample_idx);
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_TRY(dense_materialize(ctx, sample_idx, 1));
    const SampleList &L = ctx->lists[sample_idx];
    if (cap < L.n_unique) return psk_fail(ctx, PSK_ERANGE, "buffer too small: %llu < %llu", (unsigned long long)cap,
                                          (unsigned long long)L.n_unique);
    if (L.n_unique) {
        if (words) PSK_HIP(ctx, hipMemcpy(words, L.words, L.n_unique * 8, hipMemcpyDeviceToHost));
        if (freqs) PSK_HIP(ctx, hipMemcpy(freqs, L.freqs, L.n_unique * 4, hipMemcpyDeviceToHost));
    }
    return PSK_OK;
}

namespace {

// thread (i, b): lower bound of bounds[b] in the sorted list of sample i
struct SplitRef { const uint64_t *words; uint64_t n; };
__global__ void lists_split_kernel(const SplitRef *__restrict__ refs, int n, const uint64_t *__restrict__ bounds, int n_bounds,
                                   uint64_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_bounds) return;
    const int i = t / n_bounds, b = t % n_bounds;
    const uint64_t key = bounds[b];
    uint64_t lo = 0, hi = refs[i].n;
    if (b > 0 && key == 0) lo = hi;  // "end of the word space"
    const uint64_t *w = refs[i].words;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (w[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[t] = lo;
}

// 1 when words[0 .. n) ascend strictly and stay inside [lo, hi) (hi == 0: unbounded)
__global__ void list_check_kernel(const uint64_t *__restrict__ words, uint64_t n, uint64_t lo, uint64_t hi, uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t w = words[i];
    if ((i > 0 && words[i - 1] >= w) || w < lo || (hi && w >= hi)) atomicOr(bad, 1u);
}

}  // namespace

extern "C" int psk_lists_split(psk_ctx *ctx, int first_sample_idx, int n, const uint64_t *bounds, int n_bounds,
                               uint64_t *offsets_out)
{
    if (!ctx) return PSK_EINVAL;
    if (n < 0 || first_sample_idx < 0 || first_sample_idx + n > ctx->n_samples)
        return psk_fail(ctx, PSK_EINVAL, "sample range out of bounds");
    if (n == 0 || n_bounds == 0) return PSK_OK;
    if (!bounds || !offsets_out || n_bounds < 0) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    std::vector<SplitRef> refs(n);
    for (int i = 0; i < n; i++)
        if (!ctx->lists[first_sample_idx + i].done)
            return psk_fail(ctx, PSK_ESTATE, "sample %d has not been counted", first_sample_idx + i);
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_TRY(dense_materialize(ctx, first_sample_idx, n));
    for (int i = 0; i < n; i++) {
        const SampleList &L = ctx->lists[first_sample_idx + i];
        refs[i].words = L.words;
        refs[i].n = L.n_unique;
    }
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)n * n_bounds;
    const size_t b_refs = (size_t)n * sizeof(SplitRef), b_bounds = (size_t)n_bounds * 8;
    PSK_TRY(dev_reserve(ctx, ctx->flags, b_refs + b_bounds + cells * 8));
    uint8_t *base = ctx->flags.as<uint8_t>();
    PSK_HIP(ctx, hipMemcpyAsync(base, refs.data(), b_refs, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(base + b_refs, bounds, b_bounds, hipMemcpyHostToDevice, ctx->stream));
    uint64_t *d_out = reinterpret_cast<uint64_t *>(base + b_refs + b_bounds);
    lists_split_kernel<<<div_up(cells, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const SplitRef *>(base), n,
                                                                  reinterpret_cast<const uint64_t *>(base + b_refs), n_bounds, d_out);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(offsets_out, d_out, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

extern "C" int psk_copy_list_ranges(psk_ctx *ctx, int n_ranges, const int32_t *sample_idx, const uint64_t *start,
                                    const uint64_t *count, void *device_words_dst, void *device_freqs_dst)
{
    if (!ctx) return PSK_EINVAL;
    if (n_ranges < 0) return psk_fail(ctx, PSK_EINVAL, "negative range count");
    if (n_ranges == 0) return PSK_OK;
    if (!sample_idx || !start || !count) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    uint64_t total = 0;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    for (int r = 0; r < n_ranges; r++) {
        const int si = sample_idx[r];
        if (si < 0 || si >= ctx->n_samples || !ctx->lists[si].done)
            return psk_fail(ctx, PSK_ESTATE, "sample %d has not been counted", si);
        PSK_TRY(dense_materialize(ctx, si, 1));
        const SampleList &L = ctx->lists[si];
        if (start[r] > L.n_unique || count[r] > L.n_unique - start[r])
            return psk_fail(ctx, PSK_ERANGE, "range %d lies outside the list of sample %d", r, si);
        total += count[r];
    }
    if (total == 0) return PSK_OK;
    if (!device_words_dst || !device_freqs_dst) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t *dw = static_cast<uint64_t *>(device_words_dst);
    uint32_t *df = static_cast<uint32_t *>(device_freqs_dst);
    for (int r = 0; r < n_ranges; r++) {
        if (count[r] == 0) continue;
        const SampleList &L = ctx->lists[sample_idx[r]];
        PSK_HIP(ctx, hipMemcpyAsync(dw, L.words + start[r], count[r] * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(df, L.freqs + start[r], count[r] * 4, hipMemcpyDeviceToDevice, ctx->stream));
        dw += count[r];
        df += count[r];
    }
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

extern "C" int psk_release_lists(psk_ctx *ctx)
{
    if (!ctx) return PSK_EINVAL;
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t st : {ctx->copy_stream, ctx->copy_more[0], ctx->copy_more[1], ctx->copy_more[2], ctx->frame_stream, ctx->sketch_stream})
        if (st) PSK_HIP(ctx, hipStreamSynchronize(st));
    reset_lists(ctx, ctx->n_samples);   // every sample is "not counted" again
    arena_release(ctx);                 // and the chunks go back to the device, not to the next run
    ctx->have_presence = false;
    return PSK_OK;
}

extern "C" int psk_set_lists_device(psk_ctx *ctx, int n_lists, const int32_t *sample_idx, const uint64_t *count,
                                    const uint64_t *n_total, const void *device_words, const void *device_freqs)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->k == 0) return psk_fail(ctx, PSK_ESTATE, "psk_begin has not been called");
    if (n_lists < 0) return psk_fail(ctx, PSK_EINVAL, "negative list count");
    if (n_lists == 0) return PSK_OK;
    if (!sample_idx || !count) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    uint64_t total = 0;
    for (int r = 0; r < n_lists; r++) {
        if (sample_idx[r] < 0 || sample_idx[r] >= ctx->n_samples) return psk_fail(ctx, PSK_EINVAL, "sample index out of range");
        if (count[r] >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "list with more than 2^32 entries");
        total += count[r];
    }
    if (total && (!device_words || !device_freqs)) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->have_presence = false;
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *bad = ctx->misc.as<uint32_t>() + 12;
    PSK_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
    const uint64_t *sw = static_cast<const uint64_t *>(device_words);
    const uint32_t *sf = static_cast<const uint32_t *>(device_freqs);
    for (int r = 0; r < n_lists; r++) {
        SampleList &S = ctx->lists[sample_idx[r]];
        S = SampleList();
        const uint64_t n = count[r];
        if (n) {
            list_check_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(sw, n, ctx->slab_lo, ctx->slab_hi, bad);
            PSK_HIP(ctx, hipGetLastError());
            PSK_TRY(arena_alloc(ctx, n * 8, (void **)&S.words));
            PSK_TRY(arena_alloc(ctx, n * 4, (void **)&S.freqs));
            PSK_HIP(ctx, hipMemcpyAsync(S.words, sw, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(S.freqs, sf, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            sw += n;
            sf += n;
        }
        S.n_unique = n;
        S.n_total = n_total ? n_total[r] : 0;
        S.done = true;
    }
    uint32_t h_bad = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_bad) {
        for (int r = 0; r < n_lists; r++) ctx->lists[sample_idx[r]] = SampleList();
        return psk_fail(ctx, PSK_EINVAL, "a list does not ascend inside the context's slab");
    }
    return PSK_OK;
}

extern "C" int psk_lookup_counts(psk_ctx *ctx, int sample_idx, const uint64_t *words, uint64_t n, uint32_t *freqs)
{
    if (!ctx) return PSK_EINVAL;
    if (sample_idx < 0 || sample_idx >= ctx->n_samples || !ctx->lists[sample_idx].done)
        return psk_fail(ctx, PSK_ESTATE, "sample %d has not been counted", sample_idx);
    if (n == 0) return PSK_OK;
    if (!words || !freqs) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const SampleList &L = ctx->lists[sample_idx];
    PSK_TRY(dev_reserve(ctx, ctx->flags, n * 8));
    PSK_TRY(dev_reserve(ctx, ctx->starts, n * 4));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->flags.p, words, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (L.dense && !L.words)
        PSK_TRY(dense_lookup_counts(ctx, L, ctx->flags.as<uint64_t>(), n, ctx->starts.as<uint32_t>()));
    else
        lookup_counts_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(L.words, L.freqs, L.n_unique, ctx->flags.as<uint64_t>(),
                                                                      n, ctx->starts.as<uint32_t>());
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(freqs, ctx->starts.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

namespace {

// host-built open-addressing table of the dictionary on the device: slot words | slot indices; duplicate dictionary
// words share the first entry's slot (alias[d] = that entry)
struct DictTable {
    uint32_t n_slots = 0;
    uint64_t *d_words = nullptr;
    int32_t *d_idx = nullptr;
    std::vector<int32_t> alias;
};

int build_dict_table(psk_ctx *ctx, const uint64_t *dict_words, uint64_t n_dict, DictTable &T)
{
    uint32_t n_slots = DICT_SLOTS;
    while ((uint64_t)n_slots < 2 * n_dict) n_slots *= 2;
    if (n_dict >= (1ull << 30)) return psk_fail(ctx, PSK_ERANGE, "dictionary of %llu k-mers", (unsigned long long)n_dict);
    std::vector<uint64_t> sw(n_slots, 0);
    std::vector<int32_t> si(n_slots, -1);
    T.alias.assign(n_dict, -1);
    for (uint64_t d = 0; d < n_dict; d++) {
        const uint64_t w = dict_words[d];
        uint32_t h = (uint32_t)((w * 0x9E3779B97F4A7C15ull) >> 40) & (n_slots - 1);
        while (si[h] >= 0 && sw[h] != w) h = (h + 1) & (n_slots - 1);
        if (si[h] >= 0) T.alias[d] = si[h];
        else { sw[h] = w; si[h] = (int32_t)d; }
    }
    PSK_TRY(dev_reserve(ctx, ctx->flags, (size_t)n_slots * 12));
    T.n_slots = n_slots;
    T.d_words = ctx->flags.as<uint64_t>();
    T.d_idx = reinterpret_cast<int32_t *>(ctx->flags.as<uint8_t>() + (size_t)n_slots * 8);
    PSK_HIP(ctx, hipMemcpyAsync(T.d_words, sw.data(), (size_t)n_slots * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(T.d_idx, si.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // sw / si are locals
    return PSK_OK;
}

void launch_dict_count(psk_ctx *ctx, const DictTable &T, const uint8_t *d_clean, uint64_t clean_len, int k, uint32_t *d_cnt)
{
    const uint64_t threads = (clean_len + EX_SEG - 1) / EX_SEG;
    if (T.n_slots == DICT_SLOTS)
        dict_count_kernel<true><<<div_up(threads, EX_THREADS), EX_THREADS, 0, ctx->stream>>>(d_clean, clean_len, k, T.d_words, T.d_idx,
                                                                                          T.n_slots - 1, d_cnt);
    else
        dict_count_kernel<false><<<div_up(threads, EX_THREADS), EX_THREADS, 0, ctx->stream>>>(d_clean, clean_len, k, T.d_words, T.d_idx,
                                                                                           T.n_slots - 1, d_cnt);
}

// n samples (file images, or paths of uncompressed files read by the framing threads) against one dictionary:
// n_threads host threads frame ahead into the pinned ring, uploads alternate between the two buffer sets on the copy
// stream, one kernel per sample, one read-back of counts_out[n][n_dict] at the end.
int count_dict_impl(psk_ctx *ctx, int n, const uint8_t *const *bytes, const char *const *paths, const size_t *lens, int k,
                    const uint64_t *dict_words, uint64_t n_dict, uint32_t *counts_out, int n_threads)
{
    if (!ctx) return PSK_EINVAL;
    if (k < 1 || k > 32) return psk_fail(ctx, PSK_EINVAL, "k must be 1..32");
    if (n < 0) return psk_fail(ctx, PSK_EINVAL, "negative sample count");
    if (n == 0 || n_dict == 0) return PSK_OK;
    if (!dict_words || !counts_out || (!bytes && !paths) || !lens) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    DictTable T;
    PSK_TRY(build_dict_table(ctx, dict_words, n_dict, T));
    PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)n * n_dict * 4));
    uint32_t *d_cnt = ctx->starts.as<uint32_t>();
    PSK_HIP(ctx, hipMemsetAsync(d_cnt, 0, (size_t)n * n_dict * 4, ctx->stream));
    for (int i = 0; i < n; i++)
        if (paths ? !paths[i] : (!bytes[i] && lens[i])) return psk_fail(ctx, PSK_EINVAL, "null input %d", i);
    // the ingest of the counting path (pinned ring, upload + framing on the GPU two samples ahead) with the dictionary
    // kernel in the place of the counting chain
    const StreamConsumer look_up = [&](CountLane &L, int i, uint64_t clean_len) -> int {
        PSK_HIP(ctx, hipStreamWaitEvent(ctx->stream, L.raw_ready, 0));
        launch_dict_count(ctx, T, L.raw.as<uint8_t>(), clean_len, k, d_cnt + (size_t)i * n_dict);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(L.raw_free, ctx->stream));
        L.raw_used = true;
        return PSK_OK;
    };
    PSK_TRY(count_batch_impl(ctx, 0, n, bytes, paths, lens, nullptr, nullptr, n_threads, 0, 0, 0, nullptr, nullptr, k, &look_up));
    PSK_HIP(ctx, hipMemcpy(counts_out, d_cnt, (size_t)n * n_dict * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++)
        for (uint64_t d = 0; d < n_dict; d++)
            if (T.alias[d] >= 0) counts_out[(size_t)i * n_dict + d] = counts_out[(size_t)i * n_dict + T.alias[d]];
    return PSK_OK;
}

}  // namespace

extern "C" int psk_count_dict(psk_ctx *ctx, const uint8_t *bytes, size_t len, int k, const uint64_t *dict_words,
                              uint64_t n_dict, uint32_t *counts_out)
{
    if (!bytes && len) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    static const uint8_t none = 0;
    const uint8_t *one = bytes ? bytes : &none;
    return count_dict_impl(ctx, 1, &one, nullptr, &len, k, dict_words, n_dict, counts_out, 1);
}

extern "C" int psk_count_dict_batch(psk_ctx *ctx, int n, const uint8_t *const *bytes, const size_t *lens, int k,
                                    const uint64_t *dict_words, uint64_t n_dict, uint32_t *counts_out, int n_threads)
{
    return count_dict_impl(ctx, n, bytes, nullptr, lens, k, dict_words, n_dict, counts_out, n_threads);
}

extern "C" int psk_count_dict_files(psk_ctx *ctx, int n, const char *const *paths, const size_t *sizes, int k,
                                    const uint64_t *dict_words, uint64_t n_dict, uint32_t *counts_out, int n_threads)
{
    return count_dict_impl(ctx, n, nullptr, paths, sizes, k, dict_words, n_dict, counts_out, n_threads);
}
