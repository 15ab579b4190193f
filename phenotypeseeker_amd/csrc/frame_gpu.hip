// Record framing on the GPU (the host half of a1's tokeniser, DESIGN.md "Tokeniser contract"): raw FASTA / FASTQ
// file bytes in, the clean stream of kmer_count.hip out (bases kept, window breaks as '\n', everything else
// dropped) -- so that the host only moves file bytes into pinned memory.  Replaces the sequential byte state machine
// frame_sequence_counting() for the two shapes real inputs have; anything else stays with the host machine.
//
// FASTA.  Once the first record has started the machine has two states, header and sequence, and three byte
// classes: '>' always leaves it in header, '\n' always in sequence, anything else changes nothing.  The state in
// front of a byte is therefore decided by the LAST '>' or '\n' before it -- a "last marker" scan:
//   fa_summary_kernel   per 4-KB tile: type of its last marker, bytes it emits if entered in sequence / in header
//   frame_scan_kernel   one workgroup: entry state and output offset of every tile (the summaries compose
//                       associatively), total = length of the clean stream, the '\n' padding behind it
//   fa_emit_kernel      per tile again: entry state of every thread from wave ballots, bytes compacted through LDS
// FASTQ.  Four-line records: the type of a line is its index mod 4, the index a count of the newlines before it:
//   fq_lines_kernel     newlines per tile                    -> scan -> first line index of every tile
//   fq_count_kernel     bytes emitted per tile, and the checks that make the line arithmetic equal to the state
//                       machine: every header line starts with '@', every separator line with '+'
//                       -> scan -> offsets;  fq_emit_kernel  compaction as above
//   A file that fails a check (multi-line FASTQ, blank lines between records) takes the general route (r06; until then: the host):
// FASTQ, any line structure.  What the state machine does with a LINE is decided by the line before it and by the line's first
// byte -- header -> sequence; sequence -> '+' line | sequence continued (whose first byte the machine swallows: glistmaker's
// positional reading) | an empty line; '+' -> quality; quality / skipped -> '@' header | skipped -- : every line
// start is a map over eight line kinds (24 bits), maps compose associatively, so the kind of every line is a scan of maps:
//   fqg_pass_kernel<0>  per tile: the composition of its line starts' maps       -> frame_scan_kernel (mode 2) -> kind in force at every tile's first byte
//   fqg_pass_kernel<1>  bytes emitted per tile (kinds of the threads' lines from a scan of maps inside the tile) -> scan -> offsets
//   fqg_pass_kernel<2>  compaction as above
// Unlike the host machine the kernels do not collapse runs of breaks: '\n' '\n' costs a byte, not a window.
// Byte shuffling at HBM speed (two reads, one write of the file): no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int FR_THREADS = 256;
constexpr int FR_BPT = 16;                       // bytes per thread: one 16-byte load
constexpr int FR_TILE = FR_THREADS * FR_BPT;     // 4096

enum { MK_NONE = 0, MK_GT = 1, MK_NL = 2 };

// ---- byte classes of a thread's 16 bytes as 16-bit masks, four bytes per operation (SWAR) --------------------------
// A per-byte walk with its branches cost ~2400 VALU instructions per thread and made these kernels compute-bound
// (23 + 27 us for a 5-MB file, 2 % of the HBM rate); the masks below cost ~250.
struct Bytes16 {
    uint32_t w[4];
};

// the thread's 16 bytes; positions at or beyond `len` read as 0x01 (a control byte: emits nothing, marks nothing)
__device__ __forceinline__ Bytes16 load16(const uint8_t *__restrict__ raw, uint64_t len, uint64_t pos)
{
    Bytes16 b;
    if (pos + FR_BPT <= len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + pos);
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint64_t p = pos + 4 * q + e;
                x |= (uint32_t)(p < len ? raw[p] : (uint8_t)1) << (8 * e);
            }
            b.w[q] = x;
        }
    }
    return b;
}

// 0x80 in every byte of x that is zero, 0 elsewhere (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t c) { return zero_bytes(w ^ (c * 0x01010101u)); }
// the four 0x80 flags of a dword -> bits 0..3 (byte i -> bit i)
__device__ __forceinline__ uint32_t gather4(uint32_t flags) { return (((flags >> 7) * 0x01020408u) >> 24) & 0xFu; }

struct Masks16 {
    uint32_t gt, nl, at, plus, base, ctl, valid;   // bit j: byte j is '>' / '\n' / '@' / '+' / one of ACGTUacgtu / below 32 / inside the file
};

__device__ __forceinline__ Masks16 classify(const Bytes16 &b, uint64_t pos, uint64_t len)
{
    Masks16 m = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t w = b.w[q], f = w | 0x20202020u;   // f: letters folded to lower case
        m.gt |= gather4(eq_bytes(w, '>')) << (4 * q);
        m.nl |= gather4(eq_bytes(w, '\n')) << (4 * q);
        m.at |= gather4(eq_bytes(w, '@')) << (4 * q);
        m.plus |= gather4(eq_bytes(w, '+')) << (4 * q);
        m.base |= gather4(eq_bytes(f, 'a') | eq_bytes(f, 'c') | eq_bytes(f, 'g') | eq_bytes(f, 't') | eq_bytes(f, 'u')) << (4 * q);
        m.ctl |= gather4(zero_bytes(w & 0xE0E0E0E0u)) << (4 * q);
    }
    m.valid = pos >= len ? 0u : (len - pos >= FR_BPT ? 0xFFFFu : ((1u << (uint32_t)(len - pos)) - 1u));
    return m;
}

// exclusive prefix parity over the 16 positions: bit j = xor of x's bits below j
__device__ __forceinline__ uint32_t excl_prefix_xor16(uint32_t x)
{
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8;
    return (x << 1) & 0xFFFFu;
}

// byte j of the thread's 16, j a literal after unrolling
__device__ __forceinline__ uint32_t byte_at(const Bytes16 &b, int j) { return (b.w[j >> 2] >> ((j & 3) * 8)) & 0xffu; }

// the emitted bytes (mask `emit`; `keep`: the byte itself, else a break) into the LDS stage from offset `at` on
__device__ __forceinline__ void emit16(const Bytes16 &b, uint32_t emit, uint32_t keep, uint8_t *stage, uint32_t at)
{
#pragma unroll
    for (int j = 0; j < FR_BPT; j++) {
        if ((emit >> j) & 1u) stage[at + __popc(emit & ((1u << j) - 1u))] = (uint8_t)(((keep >> j) & 1u) ? byte_at(b, j) : (uint32_t)'\n');
    }
}

// ---- FASTA ------------------------------------------------------------------------------------------------------
// In sequence state: a base emits itself, '>' and every other byte from 32 up emit a break, control bytes nothing; in
// header state only the '\n' that ends the header emits (a break).  S = the state in front of every byte (1 = header):
// the last '>' or '\n' below it decides (Kogge-Stone fill of the '>' marks through the non-marker positions), the
// entry state where there is none.
struct FaThread {
    uint32_t last;          // MK_* of the thread's last marker
    uint32_t emit[2];       // bytes emitted when the thread is entered in sequence [0] / header [1] state
    uint32_t keep[2];       // ... of those, the bases
};

__device__ __forceinline__ FaThread fa_thread(const Masks16 &m)
{
    FaThread t;
    const uint32_t M = m.gt | m.nl;
    uint32_t g = m.gt, p = ~M & 0xFFFFu;
    g |= p & (g << 1); p &= p << 1;
    g |= p & (g << 2); p &= p << 2;
    g |= p & (g << 4); p &= p << 4;
    g |= p & (g << 8);
    g &= 0xFFFFu;                                                    // state AFTER every byte at or above the first marker
    const uint32_t low = M ? ((M & (0u - M)) - 1u) : 0xFFFFu;        // positions below the first marker: the entry state
    const uint32_t brk = ~(m.base | m.ctl | m.gt) & 0xFFFFu;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t A = g | (h ? low : 0u);
        const uint32_t S = ((A << 1) | (uint32_t)h) & 0xFFFFu;
        t.emit[h] = ((~S & (m.base | m.gt | brk)) | (S & m.nl)) & 0xFFFFu;
        t.keep[h] = ~S & m.base & 0xFFFFu;
    }
    t.last = M ? (((m.gt >> (31 - __clz(M))) & 1u) ? MK_GT : MK_NL) : MK_NONE;
    return t;
}

// entry state of every thread of the workgroup (true = header) given the tile's entry state; lds: 2 * (FR_THREADS / 64)
__device__ __forceinline__ bool fa_entry_state(uint32_t last, bool tile_hdr, uint32_t *lds)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t has = __ballot(last != MK_NONE), gt = __ballot(last == MK_GT);
    // the wave's own last marker, for the waves behind it
    if (lane == 0) {
        lds[wid] = has ? 1u : 0u;
        lds[FR_THREADS / 64 + wid] = has ? (uint32_t)((gt >> (63 - __builtin_clzll(has))) & 1ull) : 0u;
    }
    __syncthreads();
    bool st = tile_hdr;
    for (int w = 0; w < wid; w++)
        if (lds[w]) st = lds[FR_THREADS / 64 + w] != 0;
    const uint64_t below = has & psk_lanemask_lt(lane);
    if (below) st = (gt >> (63 - __builtin_clzll(below))) & 1ull;
    __syncthreads();
    return st;
}

// per tile: {last marker, bytes if entered in sequence, bytes if entered in header, -}
__global__ __launch_bounds__(FR_THREADS) void fa_summary_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                                 uint4 *__restrict__ summary)
{
    __shared__ uint32_t lds[2 * (FR_THREADS / 64)];
    __shared__ uint32_t red[2][FR_THREADS / 64];
    __shared__ uint32_t wlast[FR_THREADS / 64];
    const uint64_t pos = ((uint64_t)blockIdx.x * FR_THREADS + threadIdx.x) * FR_BPT;
    const FaThread t = fa_thread(classify(load16(raw, len, pos), pos, len));
    // entered in sequence / in header: the threads up to the first marker follow the tile's entry state
    const bool st_s = fa_entry_state(t.last, false, lds);
    const bool st_h = fa_entry_state(t.last, true, lds);
    uint32_t cs = __popc(t.emit[st_s ? 1 : 0]), ch = __popc(t.emit[st_h ? 1 : 0]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cs += __shfl_xor(cs, d, 64); ch += __shfl_xor(ch, d, 64); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t has = __ballot(t.last != MK_NONE), gt = __ballot(t.last == MK_GT);
    if (lane == 0) {
        red[0][wid] = cs;
        red[1][wid] = ch;
        wlast[wid] = has ? (((gt >> (63 - __builtin_clzll(has))) & 1ull) ? MK_GT : MK_NL) : MK_NONE;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0, last = MK_NONE;
        for (int w = 0; w < FR_THREADS / 64; w++) { a += red[0][w]; b += red[1][w]; if (wlast[w] != MK_NONE) last = wlast[w]; }
        summary[blockIdx.x] = make_uint4(last, a, b, 0u);
    }
}

// One workgroup.  mode 0 (FASTA): summary[t] = {last marker, bytes if entered in sequence, bytes if entered in
// header}: writes entry[t] = {entry state, output offset}.  mode 1: summary[t].x = a count: entry[t] = {exclusive
// sum, -} (used for the newline counts and, again, for the byte counts of FASTQ tiles).  total_out[0] = the total;
// pad != nullptr: 64 '\n' behind the clean stream.  host_out (pinned) receives {total, flags[0]}.
__global__ __launch_bounds__(1024) void frame_scan_kernel(const uint4 *__restrict__ summary, uint32_t n_tiles, int mode,
                                                          uint2 *__restrict__ entry, uint8_t *__restrict__ pad,
                                                          const uint32_t *__restrict__ flags, uint64_t *__restrict__ host_out)
{
    __shared__ uint32_t lds[16];
    __shared__ uint32_t w_has[16], w_gt[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t carry = 0;
    bool carry_hdr = false;   // FASTA: the file starts at its first '>' in "sequence" state (that byte opens the header)
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint4 s = t < n_tiles ? summary[t] : make_uint4(0u, 0u, 0u, 0u);
        bool hdr = false;
        uint32_t cnt = s.x;
        if (mode == 0) {
            // entry state: last marker of the tiles before this one in the chunk, else the carry
            const uint64_t has = __ballot(s.x != MK_NONE), gt = __ballot(s.x == MK_GT);
            if (lane == 0) {
                w_has[wid] = has ? 1u : 0u;
                w_gt[wid] = has ? (uint32_t)((gt >> (63 - __builtin_clzll(has))) & 1ull) : 0u;
            }
            __syncthreads();
            hdr = carry_hdr;
            for (int w = 0; w < wid; w++)
                if (w_has[w]) hdr = w_gt[w] != 0;
            const uint64_t below = has & psk_lanemask_lt(lane);
            if (below) hdr = (gt >> (63 - __builtin_clzll(below))) & 1ull;
            bool next = carry_hdr;
            for (int w = 0; w < 16; w++)
                if (w_has[w]) next = w_gt[w] != 0;
            __syncthreads();
            carry_hdr = next;
            cnt = hdr ? s.z : s.y;
        }
        uint32_t all;
        const uint32_t ex = psk_block_excl_scan_u32<1024>(cnt, &all, lds);
        if (t < n_tiles) entry[t] = make_uint2(mode == 0 ? (hdr ? 1u : 0u) : carry + ex, carry + ex);
        carry += all;
    }
    if (pad && threadIdx.x < 64) pad[(uint64_t)carry + threadIdx.x] = '\n';
    if (threadIdx.x == 0 && host_out) { host_out[0] = carry; host_out[1] = flags ? flags[0] : 0u; }
}

// compaction of a tile's bytes: every thread stores what it emits at its exclusive offset in an LDS stage, the
// workgroup copies the stage out with consecutive threads on consecutive bytes
__device__ __forceinline__ void copy_out(const uint8_t *stage, uint32_t total, uint8_t *dst)
{
    for (uint32_t i = threadIdx.x; i < total; i += FR_THREADS) dst[i] = stage[i];
}

__global__ __launch_bounds__(FR_THREADS) void fa_emit_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                              const uint2 *__restrict__ entry, uint8_t *__restrict__ clean)
{
    __shared__ uint32_t lds[2 * (FR_THREADS / 64)];
    __shared__ uint32_t scan_lds[FR_THREADS / 64];
    __shared__ uint8_t stage[FR_TILE];
    const uint64_t pos = ((uint64_t)blockIdx.x * FR_THREADS + threadIdx.x) * FR_BPT;
    const Bytes16 b = load16(raw, len, pos);
    const FaThread t = fa_thread(classify(b, pos, len));
    const uint2 e = entry[blockIdx.x];
    const int h = fa_entry_state(t.last, e.x != 0, lds) ? 1 : 0;
    uint32_t total;
    const uint32_t off = psk_block_excl_scan_u32<FR_THREADS>(__popc(t.emit[h]), &total, scan_lds);
    emit16(b, t.emit[h], t.keep[h], stage, off);
    __syncthreads();
    copy_out(stage, total, clean + e.y);
}

// ---- FASTQ (four-line records) --------------------------------------------------------------------------------------
__global__ __launch_bounds__(FR_THREADS) void fq_lines_kernel(const uint8_t *__restrict__ raw, uint64_t len, uint4 *__restrict__ summary)
{
    __shared__ uint32_t red[FR_THREADS / 64];
    const uint64_t pos = ((uint64_t)blockIdx.x * FR_THREADS + threadIdx.x) * FR_BPT;
    const Bytes16 b = load16(raw, len, pos);
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) c += __popc(eq_bytes(b.w[q], '\n'));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < FR_THREADS / 64; w++) s += red[w];
        summary[blockIdx.x] = make_uint4(s, 0u, 0u, 0u);
    }
}

// Line types: 0 header, 1 sequence, 2 separator, 3 quality; the type of a byte is (line0 + newlines below it) mod 4,
// two bits obtained from prefix parities: bit 0 flips at every newline, bit 1 at the newlines where bit 0 was set.
// Emitted: the '\n' that ends a header (the read's window break) and, on sequence lines, bases as themselves and every
// other byte from 32 up as a break.  bad: a header line that does not start with '@', a separator line not with '+'
// (the line arithmetic then differs from the state machine).
struct FqThread {
    uint32_t emit, keep;
    bool bad;
};

__device__ __forceinline__ FqThread fq_thread(const Masks16 &m, uint32_t line0, bool first_in)
{
    const uint32_t c0 = excl_prefix_xor16(m.nl) ^ ((line0 & 1u) ? 0xFFFFu : 0u);
    const uint32_t c1 = excl_prefix_xor16(m.nl & c0) ^ ((line0 & 2u) ? 0xFFFFu : 0u);
    const uint32_t T0 = ~c1 & ~c0 & 0xFFFFu, T1 = ~c1 & c0 & 0xFFFFu, T2 = c1 & ~c0 & 0xFFFFu;
    const uint32_t F = ((m.nl << 1) | (first_in ? 1u : 0u)) & m.valid;        // first byte of a line
    const uint32_t brk = ~(m.base | m.ctl) & 0xFFFFu;
    FqThread t;
    t.bad = (F & ((T0 & ~m.at) | (T2 & ~m.plus))) != 0;
    t.emit = ((m.nl & T0) | (T1 & ~m.nl & (m.base | brk))) & m.valid;
    t.keep = T1 & m.base;
    return t;
}

template <bool EMIT>
__global__ __launch_bounds__(FR_THREADS) void fq_pass_kernel(const uint8_t *__restrict__ raw, uint64_t len,
                                                              const uint2 *__restrict__ line_entry, uint4 *__restrict__ summary,
                                                              const uint2 *__restrict__ out_entry, uint8_t *__restrict__ clean,
                                                              uint32_t *__restrict__ flags)
{
    __shared__ uint32_t scan_lds[FR_THREADS / 64];
    __shared__ uint8_t stage[EMIT ? FR_TILE : 1];   // a thread emits at most one byte per byte it reads
    const uint64_t pos = ((uint64_t)blockIdx.x * FR_THREADS + threadIdx.x) * FR_BPT;
    const Bytes16 b = load16(raw, len, pos);
    const Masks16 m = classify(b, pos, len);
    uint32_t total;
    const uint32_t line0 = line_entry[blockIdx.x].x + psk_block_excl_scan_u32<FR_THREADS>(__popc(m.nl), &total, scan_lds);
    const bool first_in = pos < len && (pos == 0 || raw[pos - 1] == '\n');
    const FqThread t = fq_thread(m, line0, first_in);
    if (!EMIT) {
        // (one atomic a wave, and none once the flag is up: in a file that is NOT four-line FASTQ nearly every thread is "bad", and
        // 40 million atomics on one word made this pass take 4.8 ms for a 0.64-GB sample instead of 0.4)
        const uint64_t any_bad = __ballot(t.bad);
        if (any_bad && (threadIdx.x & 63) == 0 && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(flags, 1u);
    }
    const uint32_t off = psk_block_excl_scan_u32<FR_THREADS>(__popc(t.emit), &total, scan_lds);
    if (!EMIT) {
        if (threadIdx.x == 0) summary[blockIdx.x] = make_uint4(total, 0u, 0u, 0u);
        return;
    }
    emit16(b, t.emit, t.keep, stage, off);
    __syncthreads();
    copy_out(stage, total, clean + out_entry[blockIdx.x].y);
}

// ---- FASTQ, any line structure (r06) -----------------------------------------------------------------------------------------
// Line kinds.  SEQ0: a sequence line read from its first byte (the line behind a header, or behind a swallowed empty line); SEQ1: a
// continued sequence line -- the machine consumes the byte behind a sequence line's '\n' (kmer_count.hip: frame_sequence_counting,
// "the byte after a sequence newline is consumed"), so its first byte emits nothing; SEQE: that byte was the '\n' of an empty line;
// SKIP: a line behind the quality line that does not start with '@' (further quality lines); SKIPE: an empty such line -- its '\n' is
// taken for the line's first byte, so the line BEHIND it is skipped whatever it starts with.
enum { LK_HDR = 0, LK_SEQ0 = 1, LK_SEQ1 = 2, LK_SEQE = 3, LK_PLUS = 4, LK_QUAL = 5, LK_SKIP = 6, LK_SKIPE = 7 };
enum { LC_AT = 0, LC_PLUS = 1, LC_NL = 2, LC_OTHER = 3 };   // classes of a line's first byte
constexpr uint32_t lk_map(int k0, int k1, int k2, int k3, int k4, int k5, int k6, int k7)
{
    return (uint32_t)k0 | ((uint32_t)k1 << 3) | ((uint32_t)k2 << 6) | ((uint32_t)k3 << 9) | ((uint32_t)k4 << 12) | ((uint32_t)k5 << 15) | ((uint32_t)k6 << 18) |
           ((uint32_t)k7 << 21);
}
constexpr uint32_t LK_IDENTITY = lk_map(0, 1, 2, 3, 4, 5, 6, 7);
constexpr uint32_t LK_FIRST = lk_map(LK_HDR, LK_HDR, LK_HDR, LK_HDR, LK_HDR, LK_HDR, LK_HDR, LK_HDR);   // the record region starts at a header
// kind of a line by the kind before it (position in the map), per class of its first byte
constexpr uint32_t LK_BY_CLASS[4] = {
    /* '@'   */ lk_map(LK_SEQ0, LK_SEQ1, LK_SEQ1, LK_SEQ0, LK_QUAL, LK_HDR, LK_HDR, LK_SKIP),
    /* '+'   */ lk_map(LK_SEQ0, LK_PLUS, LK_PLUS, LK_SEQ0, LK_QUAL, LK_SKIP, LK_SKIP, LK_SKIP),
    /* '\n'  */ lk_map(LK_SEQ0, LK_SEQE, LK_SEQE, LK_SEQ0, LK_QUAL, LK_SKIPE, LK_SKIPE, LK_SKIP),
    /* other */ lk_map(LK_SEQ0, LK_SEQ1, LK_SEQ1, LK_SEQ0, LK_QUAL, LK_SKIP, LK_SKIP, LK_SKIP)};
__device__ __forceinline__ uint32_t lk_apply(uint32_t map, uint32_t kind) { return (map >> (3u * kind)) & 7u; }
__device__ __forceinline__ uint32_t lk_then(uint32_t first, uint32_t second)   // the map "first, then second"
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r |= lk_apply(second, lk_apply(first, (uint32_t)k)) << (3 * k);
    return r;
}
__device__ __forceinline__ uint32_t lk_class_map(uint32_t cls)
{
    return cls == LC_AT ? LK_BY_CLASS[0] : cls == LC_PLUS ? LK_BY_CLASS[1] : cls == LC_NL ? LK_BY_CLASS[2] : LK_BY_CLASS[3];
}

// the thread's 16 bytes: F = first bytes of lines; the map of its line starts in order; walk(): kinds and emission
struct FqgThread {
    uint32_t F, map;
};
__device__ __forceinline__ FqgThread fqg_thread(const Masks16 &m, bool first_in, bool file_start)
{
    FqgThread t;
    t.F = ((m.nl << 1) | (first_in ? 1u : 0u)) & m.valid;
    t.map = LK_IDENTITY;
    uint32_t f = t.F;
    while (f) {
        const int j = __ffs(f) - 1;
        f &= f - 1u;
        const uint32_t cls = ((m.at >> j) & 1u) ? LC_AT : ((m.plus >> j) & 1u) ? LC_PLUS : ((m.nl >> j) & 1u) ? LC_NL : LC_OTHER;
        t.map = lk_then(t.map, (file_start && j == 0) ? LK_FIRST : lk_class_map(cls));
    }
    return t;
}
// emission of the 16 bytes given the kind in force in front of them
__device__ __forceinline__ void fqg_walk(const Masks16 &m, const FqgThread &t, bool file_start, uint32_t kind, uint32_t &emit, uint32_t &keep)
{
    emit = keep = 0;
    const uint32_t brk = ~(m.base | m.ctl) & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < FR_BPT; j++) {
        const uint32_t bit = 1u << j;
        const bool first = (t.F & bit) != 0;
        if (first) {
            const uint32_t cls = (m.at & bit) ? LC_AT : (m.plus & bit) ? LC_PLUS : (m.nl & bit) ? LC_NL : LC_OTHER;
            kind = (file_start && j == 0) ? (uint32_t)LK_HDR : lk_apply(lk_class_map(cls), kind);
        }
        if (!(m.valid & bit)) continue;
        if (kind == LK_HDR) {
            if (m.nl & bit) emit |= bit;                            // the break between reads
        } else if (kind == LK_SEQ0 || (kind == LK_SEQ1 && !first)) {
            if (m.base & bit) { emit |= bit; keep |= bit; }
            else if (brk & bit) emit |= bit;                         // (control bytes -- '\n', '\r' -- emit nothing)
        }
    }
}

// exclusive scan of the threads' maps over the workgroup (maps compose in thread order); lds: FR_THREADS / 64 words
__device__ __forceinline__ uint32_t fqg_block_excl_scan(uint32_t map, uint32_t *lds, uint32_t *all)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = map;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc = lk_then(up, inc);
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t before = LK_IDENTITY, total = LK_IDENTITY;
    for (int w = 0; w < FR_THREADS / 64; w++) {
        if (w < wid) before = lk_then(before, lds[w]);
        total = lk_then(total, lds[w]);
    }
    uint32_t ex = (uint32_t)__shfl_up((int)inc, 1, 64);
    if (lane == 0) ex = LK_IDENTITY;
    __syncthreads();
    *all = total;
    return lk_then(before, ex);
}

// PASS 0: summary[tile].x = the tile's map.  PASS 1: the bytes the tile emits (kind_entry[tile].x = kind in force at its first
// byte).  PASS 2: emission.
template <int PASS>
__global__ __launch_bounds__(FR_THREADS) void fqg_pass_kernel(const uint8_t *__restrict__ raw, uint64_t len, const uint2 *__restrict__ kind_entry,
                                                               uint4 *__restrict__ summary, const uint2 *__restrict__ out_entry,
                                                               uint8_t *__restrict__ clean)
{
    __shared__ uint32_t lds[FR_THREADS / 64];
    __shared__ uint32_t scan_lds[FR_THREADS / 64];
    __shared__ uint8_t stage[PASS == 2 ? FR_TILE : 1];
    const uint64_t pos = ((uint64_t)blockIdx.x * FR_THREADS + threadIdx.x) * FR_BPT;
    const Bytes16 b = load16(raw, len, pos);
    const Masks16 m = classify(b, pos, len);
    const bool first_in = pos < len && (pos == 0 || raw[pos - 1] == '\n');
    const FqgThread t = fqg_thread(m, first_in, pos == 0);
    uint32_t all;
    const uint32_t before = fqg_block_excl_scan(t.map, lds, &all);
    if (PASS == 0) {
        if (threadIdx.x == 0) summary[blockIdx.x] = make_uint4(all, 0u, 0u, 0u);
        return;
    }
    uint32_t emit, keep;
    fqg_walk(m, t, pos == 0, lk_apply(before, kind_entry[blockIdx.x].x), emit, keep);
    uint32_t total;
    const uint32_t off = psk_block_excl_scan_u32<FR_THREADS>(__popc(emit), &total, scan_lds);
    if (PASS == 1) {
        if (threadIdx.x == 0) summary[blockIdx.x] = make_uint4(total, 0u, 0u, 0u);
        return;
    }
    emit16(b, emit, keep, stage, off);
    __syncthreads();
    copy_out(stage, total, clean + out_entry[blockIdx.x].y);
}

// one workgroup: entry[t].x = the kind in force at tile t's first byte = the maps of the tiles before it applied in order
__global__ __launch_bounds__(1024) void fqg_scan_kernel(const uint4 *__restrict__ summary, uint32_t n_tiles, uint2 *__restrict__ entry)
{
    __shared__ uint32_t lds[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t carry = LK_IDENTITY;   // (applied to any kind: the first tile's first byte forces a header anyway)
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        uint32_t inc = t < n_tiles ? summary[t].x : LK_IDENTITY;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc = lk_then(up, inc);
        }
        if (lane == 63) lds[wid] = inc;
        __syncthreads();
        uint32_t before = carry, total = carry;
        for (int w = 0; w < 16; w++) {
            if (w < wid) before = lk_then(before, lds[w]);
            total = lk_then(total, lds[w]);
        }
        uint32_t ex = (uint32_t)__shfl_up((int)inc, 1, 64);
        if (lane == 0) ex = LK_IDENTITY;
        if (t < n_tiles) entry[t] = make_uint2(lk_apply(lk_then(before, ex), (uint32_t)LK_HDR), 0u);
        __syncthreads();
        carry = total;
    }
}

}  // namespace

// Host pre-pass of one file image (a worker thread): the input ends at its first NUL and starts at its first '>' or
// '@'; that byte decides the format.  Returns 1 FASTA, 2 FASTQ, 0 nothing to frame (no record).
int frame_probe(const uint8_t *bytes, size_t len, size_t *start, size_t *end)
{
    // `len` already stops at the first NUL when the caller has looked for it (parallel_fill); look again, cheaply, only
    // in what the record search below touches
    *end = len;
    // first '>' or '@', a block at a time: the first record starts within the first bytes of any real file, and a
    // whole-file memchr for a byte that never occurs (no '>' in a FASTQ file) costs 30 ms per 0.6 GB
    for (size_t lo = 0; lo < len; lo += 4096) {
        const size_t n = len - lo < 4096 ? len - lo : 4096;
        const uint8_t *z = static_cast<const uint8_t *>(memchr(bytes + lo, 0, n));
        const size_t m = z ? (size_t)(z - (bytes + lo)) : n;
        const uint8_t *gt = static_cast<const uint8_t *>(m ? memchr(bytes + lo, '>', m) : nullptr);
        const size_t gm = gt ? (size_t)(gt - (bytes + lo)) : m;
        const uint8_t *at = static_cast<const uint8_t *>(gm ? memchr(bytes + lo, '@', gm) : nullptr);
        if (at || gt) {
            if (!z) {   // the input ends at its first NUL, wherever that is
                const void *zz = memchr(bytes + lo + n, 0, len - lo - n);
                if (zz) *end = (size_t)(static_cast<const uint8_t *>(zz) - bytes);
            } else {
                *end = lo + m;
            }
            *start = lo + (at ? (size_t)(at - (bytes + lo)) : gm);
            return at ? 2 : 1;
        }
        if (z) { *end = lo + m; break; }
    }
    *start = *end;
    return 0;
}

// the same when the caller knows where the first NUL is (nul_at = len if there is none)
int frame_probe_known_end(const uint8_t *bytes, size_t nul_at, size_t *start, size_t *end)
{
    *end = nul_at;
    for (size_t lo = 0; lo < nul_at; lo += 4096) {
        const size_t n = nul_at - lo < 4096 ? nul_at - lo : 4096;
        const uint8_t *gt = static_cast<const uint8_t *>(memchr(bytes + lo, '>', n));
        const size_t gm = gt ? (size_t)(gt - (bytes + lo)) : n;
        const uint8_t *at = static_cast<const uint8_t *>(gm ? memchr(bytes + lo, '@', gm) : nullptr);
        if (at || gt) {
            *start = lo + (at ? (size_t)(at - (bytes + lo)) : gm);
            return at ? 2 : 1;
        }
    }
    *start = nul_at;
    return 0;
}

size_t frame_gpu_scratch_bytes(uint64_t raw_len)
{
    const uint64_t n_tiles = (raw_len + FR_TILE - 1) / FR_TILE + 1;
    return (size_t)n_tiles * (16 + 8 + 8) + 256;
}

// Queues the framing of d_raw[0 .. raw_len) (format 1 / 2 as frame_probe reports, starting at the record's first
// byte; 3: FASTQ that is not four-line FASTQ -- format 2 came back with the irregular flag -- through the scan of line kinds) on `stream`: clean stream into d_clean (capacity raw_len + 128), {clean length, irregular flag} into
// host_out (pinned) when the stream reaches that point.  scratch: frame_gpu_scratch_bytes(raw_len) device bytes.
int frame_gpu_enqueue(psk_ctx *ctx, hipStream_t stream, int format, const uint8_t *d_raw, uint64_t raw_len, uint8_t *d_clean,
                      void *scratch, uint64_t *host_out)
{
    if (raw_len >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample larger than 4 GB");
    const uint32_t n_tiles = (uint32_t)((raw_len + FR_TILE - 1) / FR_TILE);
    uint4 *summary = static_cast<uint4 *>(scratch);
    uint2 *entry_a = reinterpret_cast<uint2 *>(summary + n_tiles + 1);
    uint2 *entry_b = entry_a + n_tiles + 1;
    uint32_t *flags = reinterpret_cast<uint32_t *>(entry_b + n_tiles + 1);
    if (n_tiles == 0) {
        frame_scan_kernel<<<1, 1024, 0, stream>>>(summary, 0, 1, entry_a, d_clean, nullptr, host_out);
        PSK_HIP(ctx, hipGetLastError());
        return PSK_OK;
    }
    if (format == 1) {
        fa_summary_kernel<<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, summary);
        PSK_HIP(ctx, hipGetLastError());
        frame_scan_kernel<<<1, 1024, 0, stream>>>(summary, n_tiles, 0, entry_a, d_clean, nullptr, host_out);
        PSK_HIP(ctx, hipGetLastError());
        fa_emit_kernel<<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, entry_a, d_clean);
        PSK_HIP(ctx, hipGetLastError());
        return PSK_OK;
    }
    if (format == 3) {   // FASTQ of any line structure (records over several lines, blank lines): the scan of line kinds
        fqg_pass_kernel<0><<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, nullptr, summary, nullptr, nullptr);
        PSK_HIP(ctx, hipGetLastError());
        fqg_scan_kernel<<<1, 1024, 0, stream>>>(summary, n_tiles, entry_a);
        PSK_HIP(ctx, hipGetLastError());
        fqg_pass_kernel<1><<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, entry_a, summary, nullptr, nullptr);
        PSK_HIP(ctx, hipGetLastError());
        frame_scan_kernel<<<1, 1024, 0, stream>>>(summary, n_tiles, 1, entry_b, d_clean, nullptr, host_out);
        PSK_HIP(ctx, hipGetLastError());
        fqg_pass_kernel<2><<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, entry_a, summary, entry_b, d_clean);
        PSK_HIP(ctx, hipGetLastError());
        return PSK_OK;
    }
    PSK_HIP(ctx, hipMemsetAsync(flags, 0, 4, stream));
    fq_lines_kernel<<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, summary);
    PSK_HIP(ctx, hipGetLastError());
    frame_scan_kernel<<<1, 1024, 0, stream>>>(summary, n_tiles, 1, entry_a, nullptr, nullptr, nullptr);
    PSK_HIP(ctx, hipGetLastError());
    fq_pass_kernel<false><<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, entry_a, summary, nullptr, nullptr, flags);
    PSK_HIP(ctx, hipGetLastError());
    frame_scan_kernel<<<1, 1024, 0, stream>>>(summary, n_tiles, 1, entry_b, d_clean, flags, host_out);
    PSK_HIP(ctx, hipGetLastError());
    fq_pass_kernel<true><<<n_tiles, FR_THREADS, 0, stream>>>(d_raw, raw_len, entry_a, summary, entry_b, d_clean, flags);
    PSK_HIP(ctx, hipGetLastError());
    return PSK_OK;
}

// The clean stream as these kernels produce it, for tests of the tokeniser contract (psk.h).
extern "C" int64_t psk_frame_sequence_gpu(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap)
{
    if (!ctx) return PSK_EINVAL;
    if ((!bytes && len) || !out) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    if (hipSetDevice(ctx->device) != hipSuccess) return psk_fail(ctx, PSK_EHIP, "hipSetDevice failed");
    size_t st = 0, en = 0;
    const int format = frame_probe(bytes, len, &st, &en);
    if (format == 0) return 0;
    const uint64_t rl = en - st;
    DevBuf raw, clean, scratch;
    uint64_t *res = nullptr;
    int64_t rc = PSK_OK;
    auto fail = [&](int code, const char *what) { rc = psk_fail(ctx, code, "%s", what); };
    if (dev_reserve(ctx, raw, rl + 64) || dev_reserve(ctx, clean, rl + 128) || dev_reserve(ctx, scratch, frame_gpu_scratch_bytes(rl)))
        rc = PSK_ENOMEM;
    if (rc == PSK_OK && hipHostMalloc(reinterpret_cast<void **>(&res), 64, hipHostMallocDefault) != hipSuccess) fail(PSK_ENOMEM, "hipHostMalloc failed");
    if (rc == PSK_OK && hipMemcpyAsync(raw.p, bytes + st, rl, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) fail(PSK_EHIP, "upload failed");
    if (rc == PSK_OK) rc = frame_gpu_enqueue(ctx, ctx->stream, format, raw.as<uint8_t>(), rl, clean.as<uint8_t>(), scratch.p, res);
    if (rc == PSK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) fail(PSK_EHIP, "framing kernels failed");
    if (rc == PSK_OK && format == 2 && res[1]) {   // not four-line FASTQ: the general route (what kmer_count.hip does with such a sample)
        rc = frame_gpu_enqueue(ctx, ctx->stream, 3, raw.as<uint8_t>(), rl, clean.as<uint8_t>(), scratch.p, res);
        if (rc == PSK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) fail(PSK_EHIP, "framing kernels failed");
    }
    if (rc == PSK_OK) {
        if (res[0] > out_cap) rc = psk_fail(ctx, PSK_ERANGE, "output buffer too small");
        else if (res[0] && hipMemcpy(out, clean.p, res[0], hipMemcpyDeviceToHost) != hipSuccess) fail(PSK_EHIP, "download failed");
        else rc = (int64_t)res[0];
    }
    if (res) (void)hipHostFree(res);
    dev_release(raw);
    dev_release(clean);
    dev_release(scratch);
    return rc;
}
