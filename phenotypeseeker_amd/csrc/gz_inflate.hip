// .gz inputs (row a1's "FASTA/FASTQ(.gz)"): DEFLATE decoded on the GPU.
//
// The reference hands the path of a .gz file to glistmaker, whose zlib reader inflates it on one host thread per file
// (SURVEY.md section 2 row 9, Appendix B).  Until r05 this package did the same on a pool of host threads -- ~0.5 GB/s of
// text per thread against the ~45 GB/s a PCIe link moves, so a read set in its usual form (.fastq.gz) was bound by the
// host's inflate (missing #7).  Here the COMPRESSED image crosses PCIe (a fifth of the bytes) and is inflated
// on the device: 2 GB of .fastq.gz -> 9 GB of text in 0.18 s, upload included (tools/gz_bench.py; profiles/r05_gz_*).
//
// A DEFLATE stream is serial twice over: a Huffman code has to be decoded to know where the next one starts, and a match
// copies from the 32 KB of text before it.  The way around both is the one pugz / rapidgzip take on CPUs (Kerbiriou &
// Chikhi 2019; Knespel & Brunst 2023) -- find block starts inside the stream, decode from each with the window unknown,
// resolve the unknowns afterwards -- laid out for 64-lane waves and a quarter of a million lanes:
//   1. gz_find_kernel     the stream is cut every `chunk` bytes (16 KB or more: 65,536 chunks fill the part); one wave per
//                         cut sieves the bit offsets behind it for a dynamic-Huffman block header that parses completely
//                         (code-length code, both codes complete or what zlib accepts instead, end-of-block coded).
//   2. gz_decode_kernel<false>   one LANE per chunk decodes from its start to the block end that coincides with a later
//                         chunk's start, counting the text and the matches it would write.  (A false positive of step 1
//                         never coincides with anything: the host's walk along the links from the true start of the
//                         member drops it, and the chunk before it simply runs on.)
//   3. host               the chain of chunks of every file, their places in the text; further members (cat a.gz b.gz:
//                         one more round of step 2 per member the search did not happen on; BGZF members are found by
//                         their BSIZE fields without any search); ISIZE checked.
//   4. gz_decode_kernel<true>    the same decode, now writing: literals as 16-bit symbols where they belong, matches as
//                         records {place, length, distance}.
//      gz_copy_kernel     one wave per chunk copies its matches, 64 at a time: a source before the chunk's start becomes
//                         a marker -- 256 + its place in the unknown 32-KB window --, copies of markers copy the marker.
//   5. gz_tails_kernel    one workgroup per file walks its chunks in order and resolves the last 32 KB of each (the window
//                         of the next one, kept in LDS); gz_resolve_kernel then resolves everything else at once.
//   6. gz_crc_kernel      the CRC-32 of every member against its trailer (pieces of 4 KB, combined by multiplication
//                         modulo the CRC polynomial).
// The decoder's tables are per lane: {length, symbol} over the next 8 (5) bits of the stream in LDS -- 640 B a lane, four waves
// a CU --, the canonical limits of the longer codes in registers, their symbols in LDS / registers (the overflow in global memory).
// What one lane decodes, it decodes a thousand times slower than a host core: the device wins by numbers only.  A group of
// few small files therefore goes through zlib on the call's host threads (gz_group_on_device: an estimate of both routes' times)
// -- what glistmaker does --, and so does a member the device declines (a block that runs on for megabytes without a dynamic
// header, more members than the rounds allowed here, corrupt data: zlib then words the error).
#include "dev_utils.h"
#include "psk_internal.h"

#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t GZ_NONE = ~0ull;
constexpr int GZ_FINAL = -1, GZ_ERROR = -2, GZ_OVERRUN = -3, GZ_STOP = -5;
constexpr int GZ_WIN = 32768;

// ---- the bit stream: 32-bit aligned loads into a 64-bit buffer, at least 32 valid bits after need32() ------------
struct BitIn {
    const uint32_t *w;  // the word after `ahead`
    uint32_t ahead;     // loaded one refill early: its latency passes while the symbols before it are decoded
    uint64_t bb;
    int bc;
    __device__ __forceinline__ void init(const uint8_t *base, uint64_t bit)
    {
        w = reinterpret_cast<const uint32_t *>(base) + (bit >> 5);
        bb = *w++;
        ahead = *w++;
        const int skip = (int)(bit & 31);
        bb >>= skip;
        bc = 32 - skip;
        need32();
    }
    __device__ __forceinline__ void need32()
    {
        if (bc < 32) {
            bb |= (uint64_t)ahead << bc;
            bc += 32;
            ahead = *w++;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n)
    {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t take(int n)
    {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    __device__ __forceinline__ uint64_t pos(const uint8_t *base) const
    {
        return (uint64_t)(reinterpret_cast<const uint8_t *>(w - 1) - base) * 8 - (uint64_t)bc;
    }
    __device__ __forceinline__ bool beyond(const uint32_t *endw) const { return w - 1 > endw; }
};

// sixteen 16-bit counters in registers, addressable by a run-time index
struct Pk16 {
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __device__ __forceinline__ uint32_t get(int i) const
    {
        // (masks, not selects: a chain of selects over four values is what the compiler turns into an array in scratch memory)
        const int q = i >> 2;
        const uint64_t w = (a0 & (0 - (uint64_t)(q == 0))) | (a1 & (0 - (uint64_t)(q == 1))) | (a2 & (0 - (uint64_t)(q == 2))) | (a3 & (0 - (uint64_t)(q == 3)));
        return (uint32_t)(w >> ((i & 3) * 16)) & 0xffffu;
    }
    __device__ __forceinline__ void add(int i, uint32_t v)
    {
        const uint64_t inc = (uint64_t)v << ((i & 3) * 16);
        const int q = i >> 2;
        a0 += inc & (0 - (uint64_t)(q == 0));
        a1 += inc & (0 - (uint64_t)(q == 1));
        a2 += inc & (0 - (uint64_t)(q == 2));
        a3 += inc & (0 - (uint64_t)(q == 3));
    }
    __device__ __forceinline__ uint32_t sum_from_1() const   // counters 1..15 (no field overflows: at most 320 symbols)
    {
        const uint64_t u = (a0 & ~0xffffull) + a1 + a2 + a3;
        return (uint32_t)((u & 0xffff) + ((u >> 16) & 0xffff) + ((u >> 32) & 0xffff) + ((u >> 48) & 0xffff));
    }
};

// A canonical Huffman code as DEFLATE defines it (RFC 1951 3.2.2), held as one limit and one base per code length:
// with the next MAXL bits of the stream as a number whose most significant bit is the first bit read, the symbol has
// the smallest length l with rev < lim[l], and it is entry bas[l] + (rev >> (MAXL - l)) of the symbols sorted by
// (length, value).  lim[] never decreases, so the length is 1 + the number of limits rev has reached.
template <int MAXL>
struct Huff {
    uint32_t lim[MAXL + 1];
    int32_t step[MAXL + 1];   // step[l] = bas[l + 1] - bas[l] (step[0] = bas[1]): the base is summed up along the compares, so
                              // that every access has a constant index and the arrays stay in registers
    // 1: over-subscribed.  *complete: every bit pattern is a code.
    __device__ __forceinline__ int build(const Pk16 &cnt, bool *complete)
    {
        uint32_t first = 0, off = 0;
        int32_t prev = 0;
        int bad = 0;
#pragma unroll
        for (int l = 1; l <= MAXL; l++) {
            const uint32_t c = cnt.get(l);
            first <<= 1;
            const int32_t bas = (int32_t)off - (int32_t)first;
            step[l - 1] = bas - prev;
            prev = bas;
            first += c;
            off += c;
            if (first > (1u << l)) bad = 1;
            lim[l] = first << (MAXL - l);
        }
        *complete = first == (1u << MAXL);
        return bad;
    }
    // the code length (MAXL + 1: the bits are no code of this set); *idx: which of the sorted symbols
    __device__ __forceinline__ int decode(uint32_t rev, int *idx) const
    {
        int L = 1;
        int32_t b = step[0];
#pragma unroll
        for (int l = 1; l < MAXL; l++) {
            const bool ge = rev >= lim[l];
            L += ge ? 1 : 0;
            b += ge ? step[l] : 0;
        }
        *idx = b + (int32_t)(rev >> (MAXL - L));
        return rev >= lim[MAXL] ? MAXL + 1 : L;
    }
};

struct Precode {
    Huff<7> h;
    uint64_t lo, hi;  // its symbols in canonical order, 5 bits each (12 + 7)
    __device__ __forceinline__ int symbol(int idx) const { return (int)((idx < 12 ? lo >> (5 * idx) : hi >> (5 * (idx - 12))) & 31); }
};

// HLIT, HDIST, HCLEN and the code-length code (RFC 1951 3.2.7); 1: not a header zlib would accept
__device__ __forceinline__ int gz_read_precode(BitIn &in, int &hlit, int &hdist, Precode &pc)
{
    in.need32();
    hlit = (int)in.take(5) + 257;
    hdist = (int)in.take(5) + 1;
    const int hclen = (int)in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return 1;
    constexpr int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint64_t pl = 0;  // 3 bits per symbol of the code-length alphabet
#pragma unroll
    for (int i = 0; i < 19; i++) {
        if ((i & 7) == 0) in.need32();
        if (i < hclen) pl |= (uint64_t)in.take(3) << (3 * order[i]);
    }
    Pk16 cnt;
#pragma unroll
    for (int s = 0; s < 19; s++) cnt.add((int)(pl >> (3 * s)) & 7, 1);
    bool complete;
    if (pc.h.build(cnt, &complete) || !complete) return 1;  // (inftrees.c: an incomplete CODES set is an error)
    Pk16 offs;
    {
        uint32_t o = 0;
#pragma unroll
        for (int l = 1; l <= 7; l++) {
            offs.add(l, o);
            o += cnt.get(l);
        }
    }
    pc.lo = pc.hi = 0;
#pragma unroll
    for (int s = 0; s < 19; s++) {
        const int l = (int)(pl >> (3 * s)) & 7;
        if (l) {
            const int idx = (int)offs.get(l);
            offs.add(l, 1);
            if (idx < 12) pc.lo |= (uint64_t)s << (5 * idx);
            else pc.hi |= (uint64_t)s << (5 * (idx - 12));
        }
    }
    return 0;
}

// the hlit + hdist code lengths, run-length coded in the code-length alphabet; emit(first symbol, run, length)
template <class Emit>
__device__ __forceinline__ int gz_read_lengths(BitIn &in, const uint32_t *endw, int total, const Precode &pc, Emit &&emit)
{
    int i = 0, prev = 0;
    while (i < total) {
        if (in.beyond(endw)) return 1;
        in.need32();
        int idx;
        const int L = pc.h.decode(__brev(in.peek(7)) >> 25, &idx);
        if (L > 7) return 1;
        in.drop(L);
        const int sym = pc.symbol(idx);
        int rep, len;
        if (sym < 16) {
            rep = 1;
            len = prev = sym;
        } else if (sym == 16) {
            if (i == 0) return 1;
            rep = 3 + (int)in.take(2);
            len = prev;
        } else if (sym == 17) {
            rep = 3 + (int)in.take(3);
            len = prev = 0;
        } else {
            rep = 11 + (int)in.take(7);
            len = prev = 0;
        }
        if (i + rep > total) return 1;
        emit(i, rep, len);
        i += rep;
    }
    return 0;
}

// what inftrees.c accepts of a literal/length or distance code: complete, or one code of one bit, or (distances) none
__device__ __forceinline__ bool gz_code_acceptable(const Pk16 &cnt, bool complete, bool may_be_empty)
{
    if (complete) return true;
    const uint32_t n = cnt.sum_from_1();
    if (n == 0) return may_be_empty;
    return n == 1 && cnt.get(1) == 1;
}

// counts of the code lengths of a dynamic block; the stream is left behind the lengths.  1: not acceptable
__device__ __forceinline__ int gz_count_lengths(BitIn &in, const uint32_t *endw, int hlit, int hdist, const Precode &pc, Pk16 &lc, Pk16 &dc)
{
    bool eob = false;
    const int rc = gz_read_lengths(in, endw, hlit + hdist, pc, [&](int i, int rep, int len) {
        int nl = hlit - i;
        nl = nl < 0 ? 0 : (nl > rep ? rep : nl);
        lc.add(len, (uint32_t)nl);
        dc.add(len, (uint32_t)(rep - nl));
        if (len && i <= 256 && i + rep > 256) eob = true;
    });
    return rc || !eob;   // (inflate.c: "invalid code -- missing end-of-block")
}

// ---- step 1: where a dynamic block starts ---------------------------------------------------------------------
__device__ __forceinline__ bool gz_plausible_header(const uint8_t *comp, uint64_t bit, const uint32_t *endw)
{
    BitIn in;
    in.init(comp, bit);
    if (in.peek(3) != 4) return false;  // BFINAL = 0, BTYPE = 10 (read least significant bit first)
    in.drop(3);
    int hlit, hdist;
    Precode pc;
    if (gz_read_precode(in, hlit, hdist, pc)) return false;
    Pk16 lc, dc;
    if (gz_count_lengths(in, endw, hlit, hdist, pc, lc, dc)) return false;
    Huff<15> h;
    bool complete;
    if (h.build(lc, &complete) || !gz_code_acceptable(lc, complete, false)) return false;
    if (h.build(dc, &complete) || !gz_code_acceptable(dc, complete, true)) return false;
    return true;
}

// the first 17 bits of a dynamic block that is not the last: BFINAL = 0, BTYPE = 10 (least significant bit first), HLIT and
// HDIST in range -- one offset in nine passes.  Four consecutive offsets from one 64-bit window: a bit per offset.
__device__ __forceinline__ uint32_t gz_header_starts_ok(const uint8_t *comp, uint64_t bit)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(comp) + (bit >> 5);
    const uint64_t two = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) >> (bit & 31);   // (31 + 3 + 17 bits at most)
    uint32_t ok = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t v = (uint32_t)(two >> k);
        ok |= ((v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u) ? 1u << k : 0u;
    }
    return ok;
}

// the code-length code of such a header is complete (Kraft sum 1): one survivor in a few dozen
__device__ __forceinline__ bool gz_precode_complete(const uint8_t *comp, uint64_t bit)
{
    BitIn in;
    in.init(comp, bit + 13);
    const int hclen = (int)in.take(4) + 4;
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 19; i++) {
        if ((i & 7) == 0) in.need32();
        if (i < hclen) {
            const uint32_t l = in.take(3);
            sum += l ? 128u >> l : 0u;
        }
    }
    return sum == 128u;
}

// One wave per search: the first bit offset in [from, to) at which a dynamic block header parses.  Three sieves, each run
// on FULL waves: the offsets that pass one wait in a queue (LDS, in order) until sixty-four of them are there for the next
// -- the complete parse costs thousands of instructions and one offset in a thousand gets that far; run where it arises,
// it would be executed by one lane while sixty-three wait.
__global__ __launch_bounds__(64) void gz_find_kernel(const uint8_t *comp, const uint64_t *from, const uint64_t *to, const uint64_t *end_byte,
                                                      int n, uint64_t *found)
{
    __shared__ uint32_t q1[64 + 256], q2[128];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= n) return;
    const uint64_t b0 = from[c], b1 = to[c];
    const uint32_t *endw = reinterpret_cast<const uint32_t *>(comp + end_byte[c]) + 2;
    uint32_t n1 = 0, n2 = 0;
    uint64_t hit = GZ_NONE;
    // (the queues are one wave's: its DS instructions execute in order, so a lane reads what another wrote an instruction
    // earlier without any wait; the barrier only keeps the compiler from moving them)
    auto push = [&](uint32_t *q, uint32_t &nq, bool ok, uint32_t value) {
        const uint64_t m = __ballot(ok);
        if (ok) q[nq + __popcll(m & ((1ull << lane) - 1))] = value;
        nq += (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    };
    // the complete parse of the first 64 (or all, at the end) of q2
    auto sieve3 = [&](uint32_t take) {
        const bool have = (uint32_t)lane < take;
        const uint32_t off = have ? q2[lane] : 0;
        const bool ok = have && gz_plausible_header(comp, b0 + off, endw);
        const uint64_t m = __ballot(ok);
        if (m) hit = b0 + (uint64_t)__builtin_amdgcn_readlane((int)off, __ffsll((long long)m) - 1);
        const uint32_t rest = n2 - take;   // (the queue moves down)
        const uint32_t moved = (uint32_t)lane < rest ? q2[take + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < rest) q2[lane] = moved;
        n2 = rest;
        __builtin_amdgcn_wave_barrier();
    };
    auto sieve2 = [&](uint32_t take) {
        const bool have = (uint32_t)lane < take;
        const uint32_t off = have ? q1[lane] : 0;
        const bool ok = have && gz_precode_complete(comp, b0 + off);
        // (the queue moves down: up to 256 entries wait behind the 64 taken)
        const uint32_t rest = n1 - take;
        uint32_t moved[4];
#pragma unroll
        for (int i = 0; i < 4; i++) moved[i] = (uint32_t)lane + 64 * i < rest ? q1[take + lane + 64 * i] : 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((uint32_t)lane + 64 * i < rest) q1[lane + 64 * i] = moved[i];
        n1 = rest;
        push(q2, n2, ok, off);
    };
    for (uint64_t b = b0; b < b1 && hit == GZ_NONE; b += 256) {
        // lane l: offsets b + 4 l .. + 3; its survivors go into the queue in order, behind those of the lanes before it
        const uint64_t bit = b + 4 * lane;
        uint32_t ok = bit < b1 ? gz_header_starts_ok(comp, bit) : 0;
        if (bit + 4 > b1) ok &= bit < b1 ? (1u << (b1 - bit)) - 1u : 0u;
        const uint32_t cnt = (uint32_t)__popc(ok), upto = psk_wave_incl_scan_u32(cnt, lane);
        uint32_t at = n1 + upto - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ok & (1u << k)) q1[at++] = (uint32_t)(bit - b0) + k;
        n1 += (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
        __builtin_amdgcn_wave_barrier();
        while (n1 >= 64 && hit == GZ_NONE) {
            sieve2(64);
            // (sixteen, not sixty-four: a header that has come this far is usually the true one, and the search would run on
            // for another 64 K offsets before it is looked at)
            if (n2 >= 16) sieve3(n2 < 64 ? n2 : 64);
        }
    }
    while (hit == GZ_NONE && (n1 || n2)) {
        if (n1) sieve2(n1 < 64 ? n1 : 64);
        while (hit == GZ_NONE && n2 && (n2 >= 64 || !n1)) sieve3(n2 < 64 ? n2 : 64);
    }
    if (lane == 0) found[c] = hit;
}

// ---- steps 2 and 4: the decoder, one lane per chunk -------------------------------------------------------------
struct GzDecodeArgs {
    const uint8_t *comp;         // every file's image, 16-byte aligned, zeros between them
    const uint64_t *start_bit;   // per chunk; GZ_NONE: nothing starts in this chunk
    const uint64_t *end_byte;    // per chunk: where its file's image ends
    const uint8_t *true_start;   // per chunk: a member starts here (no text before it)
    uint16_t *long_syms;         // per chunk 320 u16: the symbols whose codes are longer than the tables' index
    // counting pass: the starts a block end may coincide with, ascending, entries [cand_from, cand_to) of cand_bit
    const uint64_t *cand_bit;
    const uint32_t *cand_from, *cand_to;
    uint64_t max_span_bits;
    // writing pass
    const uint64_t *stop_bit, *out_off, *rec_off, *want_len, *want_rec;
    uint16_t *sym;
    uint2 *rec;                  // a match: {where in the chunk's text, length | distance << 16}
    // results
    uint64_t *out_len, *n_rec, *end_bit;
    int32_t *link;               // counting: entry of cand_bit reached, or GZ_FINAL / GZ_ERROR / GZ_OVERRUN; writing: GZ_STOP / GZ_FINAL / GZ_ERROR
    int n;
};

constexpr int GZ_LIT_BITS = 8, GZ_DIST_BITS = 5, GZ_LONG_LDS = 32;
constexpr int GZ_LDS_U16 = ((1 << GZ_LIT_BITS) + (1 << GZ_DIST_BITS) + GZ_LONG_LDS) * 64;   // 40 KB a wave: four waves a CU

// The codes of a block, per lane: a table over the next 8 (5) bits of the stream in LDS -- {code length, symbol}, 0 for the
// prefix of a longer code -- and, for the longer codes, the canonical limits in registers and the sorted symbols: the first
// 32 literal/length ones in LDS, the distance ones packed into three registers, the rest (rare) in global memory.  A global
// load in the symbol loop stalls the WAVE for a memory latency whenever any of its 64 lanes takes it.
struct LongCodes {
    uint32_t lim[16];   // only the entries above the table's index width are used
    int32_t step[16];
    int32_t base;       // of the first length the table does not cover
    template <int FROM>
    __device__ __forceinline__ void take(const Huff<15> &h)
    {
        int32_t b = 0;
#pragma unroll
        for (int l = 0; l < FROM; l++) b += h.step[l];
        base = b;
#pragma unroll
        for (int l = FROM; l <= 15; l++) {
            lim[l] = h.lim[l];
            step[l] = h.step[l];
        }
    }
    // rev: the next 15 bits, first bit most significant, known not to start a code shorter than FROM
    template <int FROM>
    __device__ __forceinline__ int decode(uint32_t rev, int *idx) const
    {
        int L = FROM;
        int32_t b = base;
#pragma unroll
        for (int l = FROM; l < 15; l++) {
            const bool ge = rev >= lim[l];
            L += ge ? 1 : 0;
            b += ge ? step[l] : 0;
        }
        *idx = b + (int32_t)(rev >> (15 - L));
        return rev >= lim[15] ? 16 : L;
    }
};

template <bool WRITE>
__global__ __launch_bounds__(64) void gz_decode_kernel(GzDecodeArgs a)
{
    extern __shared__ uint16_t gz_lds[];
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    if (c >= a.n) return;
    const uint64_t start = a.start_bit[c];
    if (start == GZ_NONE) return;
    uint16_t *lt = gz_lds + lane;                               // entry t of this lane: lt[t * 64]
    uint16_t *dt = gz_lds + (1 << GZ_LIT_BITS) * 64 + lane;
    uint16_t *ls = gz_lds + ((1 << GZ_LIT_BITS) + (1 << GZ_DIST_BITS)) * 64 + lane;   // long literal/length symbol i: ls[i * 64]
    uint16_t *lsym = a.long_syms + (size_t)c * 320;
    uint64_t dpk0 = 0, dpk1 = 0, dpk2 = 0;   // the long distance symbols, 5 bits each, twelve a register
    uint32_t l_short = 0, d_short = 0;       // how many symbols the tables hold themselves
    const uint8_t *comp = a.comp;
    const uint32_t *endw = reinterpret_cast<const uint32_t *>(comp + a.end_byte[c]) + 2;
    const bool true_start = a.true_start[c] != 0;
    uint32_t nc = WRITE ? 0 : a.cand_from[c];
    const uint32_t nc_end = WRITE ? 0 : a.cand_to[c];
    const uint64_t stop = WRITE ? a.stop_bit[c] : 0;
    uint16_t *out = WRITE ? a.sym + a.out_off[c] : nullptr;
    uint2 *rec = WRITE ? a.rec + a.rec_off[c] : nullptr;
    // (the writing pass never writes beyond what the counting pass counted, whatever it decodes)
    const uint64_t room_text = WRITE ? a.want_len[c] : ~0ull, room_rec = WRITE ? a.want_rec[c] : ~0ull;

    BitIn in;
    in.init(comp, start);
    LongCodes ll, ld;
    uint64_t pos = 0, nrec = 0;
    int link = GZ_ERROR;
    bool final = false;
    for (;;) {  // blocks
        const uint64_t bit = in.pos(comp);
        if (final) {
            link = GZ_FINAL;
            break;
        }
        if (WRITE) {
            if (bit >= stop) {
                link = bit == stop ? GZ_STOP : GZ_ERROR;
                break;
            }
        } else {
            while (nc < nc_end && a.cand_bit[nc] < bit) nc++;
            if (nc < nc_end && a.cand_bit[nc] == bit) {
                link = (int)nc;
                break;
            }
            if (bit - start > a.max_span_bits) {
                link = GZ_OVERRUN;
                break;
            }
        }
        if (in.beyond(endw) || pos >= (1ull << 32)) break;
        in.need32();
        final = in.take(1) != 0;
        const uint32_t type = in.take(2);
        if (type == 3) break;
        if (type == 0) {  // stored
            in.drop(in.bc & 7);
            in.need32();
            const uint32_t len = in.take(16);
            in.need32();
            const uint32_t nlen = in.take(16);
            if ((len ^ nlen) != 0xffffu) break;
            bool bad = false;
            for (uint32_t j = 0; j < len; j++) {
                if (in.beyond(endw)) {
                    bad = true;
                    break;
                }
                in.need32();
                const uint32_t b = in.take(8);
                if (WRITE && pos >= room_text) {
                    bad = true;
                    break;
                }
                if (WRITE) out[pos] = (uint16_t)b;
                pos++;
            }
            if (bad) break;
            continue;
        }
        // ---- the block's two codes ----
        {
            Pk16 lc, dc;
            int hlit = 288, hdist = 32;
            Precode pc;
            BitIn lengths_at = in;
            if (type == 1) {  // the fixed code (RFC 1951 3.2.6)
                lc.add(7, 24);
                lc.add(8, 152);
                lc.add(9, 112);
                dc.add(5, 32);
            } else {
                if (gz_read_precode(in, hlit, hdist, pc)) break;
                lengths_at = in;
                if (gz_count_lengths(in, endw, hlit, hdist, pc, lc, dc)) break;
            }
            Huff<15> h;
            bool lcomplete, dcomplete;
            if (h.build(lc, &lcomplete) || !gz_code_acceptable(lc, lcomplete, false)) break;
            ll.take<GZ_LIT_BITS + 1>(h);
            Pk16 lcode, loff, dcode, doff;   // per length: the next code, the next place among the sorted symbols
            {
                uint32_t first = 0, o = 0;
#pragma unroll
                for (int l = 1; l <= 15; l++) {
                    first <<= 1;
                    lcode.add(l, first);
                    loff.add(l, o);
                    first += lc.get(l);
                    o += lc.get(l);
                }
            }
            if (h.build(dc, &dcomplete) || !gz_code_acceptable(dc, dcomplete, true)) break;
            ld.take<GZ_DIST_BITS + 1>(h);
            {
                uint32_t first = 0, o = 0;
#pragma unroll
                for (int l = 1; l <= 15; l++) {
                    first <<= 1;
                    dcode.add(l, first);
                    doff.add(l, o);
                    first += dc.get(l);
                    o += dc.get(l);
                }
            }
            l_short = d_short = 0;
#pragma unroll
            for (int l = 1; l <= 15; l++) {
                if (l <= GZ_LIT_BITS) l_short += lc.get(l);
                if (l <= GZ_DIST_BITS) d_short += dc.get(l);
            }
            dpk0 = dpk1 = dpk2 = 0;
            // an incomplete code leaves bit patterns that are no code: they must not find an entry of the block before
            if (!lcomplete)
                for (int t = 0; t < (1 << GZ_LIT_BITS); t++) lt[t * 64] = 0;
            if (!dcomplete)
                for (int t = 0; t < (1 << GZ_DIST_BITS); t++) dt[t * 64] = 0;
            auto place = [&](int i, int rep, int len) {
                if (!len) return;
                for (int j = i; j < i + rep; j++) {
                    const bool is_lit = j < hlit;
                    const uint32_t code = is_lit ? lcode.get(len) : dcode.get(len);
                    const uint32_t at = is_lit ? loff.get(len) : doff.get(len);
                    if (is_lit) {
                        lcode.add(len, 1);
                        loff.add(len, 1);
                    } else {
                        dcode.add(len, 1);
                        doff.add(len, 1);
                    }
                    const uint32_t r = __brev(code) >> (32 - len);   // the code as the stream presents it: first bit lowest
                    const int bits = is_lit ? GZ_LIT_BITS : GZ_DIST_BITS;
                    uint16_t *tab = is_lit ? lt : dt;
                    const int value = is_lit ? j : j - hlit;
                    if (len <= bits) {
                        const uint16_t e = (uint16_t)((value << 4) | len);
                        for (uint32_t t = r; t < (1u << bits); t += 1u << len) tab[t * 64] = e;
                    } else {
                        tab[(r & ((1u << bits) - 1)) * 64] = 0;
                        if (is_lit) {
                            const uint32_t rel = at - l_short;
                            if (rel < GZ_LONG_LDS) ls[rel * 64] = (uint16_t)value;
                            else lsym[at] = (uint16_t)value;
                        } else {
                            const uint32_t rel = at - d_short, q = rel / 12, sh = 5 * (rel % 12);
                            const uint64_t bits5 = (uint64_t)value << sh;
                            dpk0 |= q == 0 ? bits5 : 0;
                            dpk1 |= q == 1 ? bits5 : 0;
                            dpk2 |= q == 2 ? bits5 : 0;
                        }
                    }
                }
            };
            if (type == 1) {
                place(0, 144, 8);
                place(144, 112, 9);
                place(256, 24, 7);
                place(280, 8, 8);
                place(288, 32, 5);
            } else {
                in = lengths_at;
                (void)gz_read_lengths(in, endw, hlit + hdist, pc, place);
            }
        }
        // ---- the block's symbols ----
        // (one way out of the loop: an error only raises a flag and lets the iteration finish on harmless values -- every `break`
        // of its own costs the wave a round of exec-mask bookkeeping in EVERY iteration)
        bool bad = false;
        for (;;) {
            bool wrong = in.beyond(endw);
            in.need32();
            uint32_t e = lt[in.peek(GZ_LIT_BITS) * 64];
            int len = (int)(e & 15);
            uint32_t s = e >> 4;
            if (len == 0) {
                int idx;
                len = ll.decode<GZ_LIT_BITS + 1>(__brev(in.peek(15)) >> 17, &idx);
                const bool none = len > 15;
                wrong = wrong || none;
                len = none ? 1 : len;
                const uint32_t rel = none ? 0 : (uint32_t)idx - l_short;
                s = ls[(rel < GZ_LONG_LDS ? rel : 0) * 64];
                if (rel >= GZ_LONG_LDS) {
                    // (rare.  The wait is HERE so that the common path carries no pending global load of a symbol: a wait for
                    // one where the paths join would also wait, in every iteration, for the input word that was just asked for)
                    s = lsym[idx];
                    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
                }
            }
            in.drop(len);
            if (s < 256) {
                const bool room = !WRITE || pos < room_text;
                if (WRITE && room && !wrong) out[pos] = (uint16_t)s;
                wrong = wrong || !room;
                pos++;
            } else if (s == 256) {
                bad = wrong;   // the end of the block -- unless something before it was wrong
                break;
            } else {
                const int i = (int)s - 257;
                wrong = wrong || i > 28;
                const int eb = i < 8 || i >= 28 ? 0 : (i - 4) >> 2;
                const uint32_t mlen = (i < 8 ? 3u + i : i >= 28 ? 258u : 3u + ((4u + (i & 3)) << eb)) + in.take(eb);
                in.need32();
                e = dt[in.peek(GZ_DIST_BITS) * 64];
                len = (int)(e & 15);
                uint32_t ds = e >> 4;
                if (len == 0) {
                    int idx;
                    len = ld.decode<GZ_DIST_BITS + 1>(__brev(in.peek(15)) >> 17, &idx);
                    const bool none = len > 15;
                    wrong = wrong || none;
                    len = none ? 1 : len;
                    const uint32_t rel = none ? 0 : (uint32_t)idx - d_short, q = rel / 12;
                    ds = (uint32_t)((q == 0 ? dpk0 : q == 1 ? dpk1 : dpk2) >> (5 * (rel % 12))) & 31;
                }
                in.drop(len);
                wrong = wrong || ds >= 30;
                ds = ds >= 30 ? 0 : ds;
                const int db = ds < 4 ? 0 : (int)(ds >> 1) - 1;
                const uint32_t dist = (ds < 4 ? ds + 1 : 1u + ((2u + (ds & 1)) << db)) + in.take(db);
                wrong = wrong || (true_start && dist > pos) || ((pos + mlen) >> 32);   // (inflate.c: "invalid distance too far back")
                if (WRITE) {
                    const bool room = nrec < room_rec && pos + mlen <= room_text;
                    if (room && !wrong) rec[nrec] = make_uint2((uint32_t)pos, mlen | (dist << 16));
                    wrong = wrong || !room;
                }
                nrec++;
                pos += mlen;
            }
            if (wrong) {
                bad = true;
                break;
            }
        }
        if (bad) break;
    }
    a.out_len[c] = pos;
    a.n_rec[c] = nrec;
    a.end_bit[c] = in.pos(comp);
    if (WRITE && link != GZ_ERROR && (pos != a.want_len[c] || nrec != a.want_rec[c])) link = GZ_ERROR;
    a.link[c] = link;
}

// ---- step 4b: the matches -----------------------------------------------------------------------------------------
// One wave per chunk, 64 matches at a time, one per lane: a match whose source holds nothing that an earlier match of the
// same 64 writes is copied at once (sixty-four loads in flight instead of one), the others in the rounds after the ones
// they wait for.  A source that lies before the chunk is a marker: 256 + its place in the unknown window.
// (r06: the same copies through a ring of the chunk's text in LDS -- one wave a chunk, 76 KB, literals streamed in and text streamed
// out in 16-byte pieces: every symbol once in, once out instead of 4.7 x -- were correct (all of tests/test_gpu_gz*.py) and SEVEN
// TIMES SLOWER, 344 ms against 48: two chunks a CU instead of twenty, and nothing hides the LDS round trip of a symbol-by-symbol
// copy of up to 258 symbols.  What this kernel is bound by is waves in flight, not bytes.  docs/NOTEBOOK.md, round 6.)
constexpr int GZ_COPY_WIDE = 2;   // 16-byte pieces a lane has in flight

// The records are the writing pass's: a chunk whose pass ended in GZ_ERROR has left its records (and symbols) half written --
// what lies behind them is an earlier group's data or uninitialised memory: such a chunk is not copied at all, no
// chunk copies more records than its pass wrote, and a record that would write outside its chunk's text is dropped whatever
// wrote it (its file then fails the check sum and goes to zlib).
// (r06, same question from the other side: forcing the register budget down to 6 / 8 waves a SIMD -- 80 / 64 VGPRs with 64 / 128 bytes
// of scratch a lane -- made it slower too: 48.6 -> 51.8 / 57.1 ms.)
__global__ __launch_bounds__(256) void gz_copy_kernel(uint16_t *sym, const uint2 *rec, const uint64_t *rec_off, const uint64_t *want_rec,
                                                       const uint64_t *got_rec, const int32_t *link, const uint64_t *want_len,
                                                       const uint64_t *out_off, int n_chunks, unsigned long long *stats)
{
    __shared__ uint32_t g_d[4][64], g_e[4][64], g_ld[4][64];   // the 64 matches in hand: first and last + 1 symbol written; length | distance << 16
    __shared__ long long g_s[4][64];                           // where they read (after the redirections below)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wv;
    if (c >= n_chunks || link[c] == GZ_ERROR) return;
    uint16_t *out = sym + out_off[c];
    const uint2 *r = rec + rec_off[c];
    const uint64_t n = got_rec[c] < want_rec[c] ? got_rec[c] : want_rec[c];
    const uint64_t text_len = want_len[c];
    auto wave_sync = [] {
        // the readers are lanes of this very wave: the stores only have to have left it (a fence of agent scope writes the
        // L2 back on a part with eight of them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const uint2 none = make_uint2(0xffffffffu, 0);
    uint2 x = (uint64_t)lane < n ? r[lane] : none;
    for (uint64_t g = 0; g < n; g += 64) {
        const uint2 nx = g + 64 + lane < n ? r[g + 64 + lane] : none;   // (the next 64, on their way while these are copied)
        const uint32_t d = x.x, len = x.y & 0xffffu, dist = x.y >> 16;
        const bool valid = g + lane < n && dist != 0 && dist <= (uint32_t)GZ_WIN && len != 0 && (uint64_t)d + len <= text_len;
        int64_t s = (int64_t)d - (int64_t)dist;
        const int64_t span = len < dist ? len : dist;   // (an overlapping match reads only the `dist` symbols before it)
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        uint64_t dep = 0;
        if (stats && lane == 0) atomicAdd(stats + 0, 1ull);
        if (__ballot(valid && s + span > (int64_t)d0)) {
            // Some match reads what a match of these 64 writes.  A chain of them -- the header of a read copied from the header
            // of the read before, that one from the one before it -- would be copied link by link, a memory round trip each;
            // but a match that reads INSIDE an earlier (non-overlapping) match reads what that one read: it is pointed there,
            // again and again, until its source is text that is already in place.
            if (stats && lane == 0) atomicAdd(stats + 2, 1ull);
            g_d[wv][lane] = valid ? d : 0xffffffffu;   // (a dropped record is nobody's source)
            g_e[wv][lane] = valid ? d + len : 0xffffffffu;
            g_ld[wv][lane] = x.y;
            g_s[wv][lane] = s;
            __builtin_amdgcn_wave_barrier();   // (LDS of one wave: its DS instructions execute in order; nothing to wait for)
            // the first earlier match that ends behind `from` (the matches write ascending, disjoint ranges)
            auto first_ending_behind = [&](int64_t from) {
                int lo = -1, hi = lane;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if ((int64_t)g_e[wv][mid] > from) hi = mid;
                    else lo = mid;
                }
                return hi;   // lane: none
            };
            for (int hop = 0; hop < 64; hop++) {
                bool moved = false;
                if (valid && s >= (int64_t)d0) {
                    const int i = first_ending_behind(s);
                    if (i < lane) {
                        const uint32_t di = g_d[wv][i], ldi = g_ld[wv][i], li = ldi & 0xffffu, disti = ldi >> 16;
                        if ((int64_t)di <= s && s + span <= (int64_t)di + li && disti >= li) {
                            s = g_s[wv][i] + (s - (int64_t)di);
                            moved = true;
                        }
                    }
                }
                if (!__ballot(moved)) break;
                if (stats && lane == 0) atomicAdd(stats + 3, 1ull);
                if (moved) g_s[wv][lane] = s;   // (later matches pointed at this one follow it)
                __builtin_amdgcn_wave_barrier();
            }
            // what is left: a match that reads part of what earlier ones write waits for them -- they are neighbours
            const int64_t e = s + span;
            if (valid && e > (int64_t)d0) {
                const int i0 = first_ending_behind(s);
                if (i0 < lane && (int64_t)g_d[wv][i0] < e) {
                    int lo = i0, hi = lane;   // the last earlier match that starts before e
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if ((int64_t)g_d[wv][mid] < e) lo = mid;
                        else hi = mid;
                    }
                    dep = ((lo == 63 ? 0ull : (1ull << (lo + 1))) - 1ull) & ~((1ull << i0) - 1ull);
                }
            }
        }
        uint64_t done = ~__ballot(valid);
        if (stats) {
            const uint64_t dm = __ballot(dep != 0);
            if (lane == 0) atomicAdd(stats + 5, (unsigned long long)__popcll(dm));
        }
        // mode of the copy: 0 sixteen bytes a request (the addresses are two-byte aligned only: gfx950 takes unaligned global
        // accesses); 1 a run whose period divides eight: one 16-byte pattern fills it; 2 symbol by symbol (markers before the
        // chunk, other overlaps)
        const int mode = s >= 0 && dist >= len ? 0 : (s >= 0 && (dist == 1 || dist == 2 || dist == 4) ? 1 : 2);
        while (~done) {
            const bool ready = !((done >> lane) & 1) && (dep & ~done) == 0;
            if (stats && lane == 0) atomicAdd(stats + 1, 1ull);
            uint16_t *dst = out + d;
            const uint16_t *src = out + s;
            // ---- loads of everything this round copies in 16-byte pieces, then the stores: a load queued behind a store waits
            // for the store's acknowledgement as well (one counter for both on this part)
            uint32_t longest = ready && mode != 2 ? len : 0;
            for (int o = 32; o; o >>= 1) {
                const uint32_t t = (uint32_t)__shfl_xor((int)longest, o, 64);
                longest = t > longest ? t : longest;
            }
            uint4 pat = make_uint4(0, 0, 0, 0);
            if (ready && mode == 1) {
                uint16_t p[4];
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = src[i % (int)dist];
                const uint32_t lo = p[0] | ((uint32_t)p[1] << 16), hi = p[2] | ((uint32_t)p[3] << 16);
                pat = make_uint4(lo, hi, lo, hi);
            }
            for (uint32_t j0 = 0; j0 < longest; j0 += 8 * GZ_COPY_WIDE) {
                const bool here = ready && mode != 2 && j0 < len;
                const uint32_t n_here = here ? (len - j0 < 8 * GZ_COPY_WIDE ? len - j0 : 8 * GZ_COPY_WIDE) : 0, nv = n_here >> 3, nt = n_here & 7;
                uint4 v[GZ_COPY_WIDE];
                uint16_t t[7];
                if (mode == 0) {
#pragma unroll
                    for (int i = 0; i < GZ_COPY_WIDE; i++)
                        if ((uint32_t)i < nv) __builtin_memcpy(&v[i], src + j0 + 8 * i, 16);
#pragma unroll
                    for (int i = 0; i < 7; i++)
                        if ((uint32_t)i < nt) t[i] = src[j0 + 8 * nv + i];
                } else {
#pragma unroll
                    for (int i = 0; i < GZ_COPY_WIDE; i++) v[i] = pat;
#pragma unroll
                    for (int i = 0; i < 7; i++) t[i] = (uint16_t)((i & 1 ? (i & 2 ? pat.y : pat.x) >> 16 : (i & 2 ? pat.y : pat.x)) & 0xffffu);
                }
#pragma unroll
                for (int i = 0; i < GZ_COPY_WIDE; i++)
                    if ((uint32_t)i < nv) __builtin_memcpy(dst + j0 + 8 * i, &v[i], 16);
#pragma unroll
                for (int i = 0; i < 7; i++)
                    if ((uint32_t)i < nt) dst[j0 + 8 * nv + i] = t[i];
            }
            // ---- the rest, eight symbols at a time; the sources wrap at `dist` (an overlapping match repeats its first
            // `dist` symbols), so every load reads text written before this match
            if (__ballot(ready && mode == 2)) {
                uint32_t k = 0;   // j mod dist
                uint32_t slow = ready && mode == 2 ? len : 0;
                for (int o = 32; o; o >>= 1) {
                    const uint32_t t = (uint32_t)__shfl_xor((int)slow, o, 64);
                    slow = t > slow ? t : slow;
                }
                for (uint32_t j = 0; j < slow; j += 8) {
                    const bool here = ready && mode == 2;
                    uint16_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int64_t at = s + k;
                        v[i] = at < 0 ? (uint16_t)(256 + GZ_WIN + at) : (here && j + i < len ? out[at] : (uint16_t)0);
                        k = k + 1 == dist ? 0 : k + 1;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (here && j + i < len) dst[j + i] = v[i];
                }
            }
            wave_sync();   // what this round wrote, the next one (and the next 64 matches) may read
            done |= __ballot(ready);
        }
        x = nx;
    }
}

// ---- step 5: markers to bytes -----------------------------------------------------------------------------------
// Both kernels also note where a file's text has its first NUL byte (the framing ends an input there: frame_gpu.hip).
// One workgroup per file, its chunks in stream order: the last 32 KB of each, whose markers point into the last 32 KB
// before the chunk -- resolved by the iterations before.
__global__ __launch_bounds__(1024) void gz_tails_kernel(const uint16_t *sym, uint8_t *out, const uint64_t *off, const uint64_t *len,
                                                         const uint32_t *file_first, const uint8_t *no_window, unsigned long long *first_nul)
{
    if (no_window[blockIdx.x]) return;   // every chunk a member of its own (BGZF: tens of thousands a file): gz_resolve_kernel does them whole
    // the window of the chunk in hand -- the 32 KB of text before it -- as a ring in LDS: position w is ring[(head + w) % 32K];
    // the symbols of the next chunk's tail are loaded while this one's are resolved (a file's chunks are a serial chain:
    // what is waited for per link is what the chain costs)
    __shared__ uint8_t ring[GZ_WIN];
    const uint32_t c0 = file_first[blockIdx.x], c1 = file_first[blockIdx.x + 1];
    const uint32_t tid = threadIdx.x;
    uint32_t head = 0;
    uint16_t cur[32], nxt[32];
    auto load = [&](uint32_t c, uint16_t (&v)[32]) {
        const uint64_t L = len[c], t = L < GZ_WIN ? L : GZ_WIN, base = off[c] + L - t;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const uint32_t i = tid + j * 1024;
            v[j] = i < t ? sym[base + i] : (uint16_t)0;
        }
    };
    if (c0 < c1) load(c0, cur);
    for (uint32_t c = c0; c < c1; c++) {
        if (c + 1 < c1) load(c + 1, nxt);
        const uint64_t L = len[c], t = L < GZ_WIN ? L : GZ_WIN, base = off[c] + L - t;
        uint8_t b[32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const uint32_t s = cur[j];
            b[j] = s < 256 ? (uint8_t)s : ring[(head + (s - 256)) & (GZ_WIN - 1)];
        }
        __syncthreads();   // every marker has been looked up: the ring may take the new text
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const uint32_t i = tid + j * 1024;
            if (i < t) {
                ring[(head + i) & (GZ_WIN - 1)] = b[j];
                out[base + i] = b[j];
                if (b[j] == 0) atomicMin(first_nul + blockIdx.x, (unsigned long long)(base + i));
            }
        }
        head = (head + (uint32_t)t) & (GZ_WIN - 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) cur[j] = nxt[j];
    }
}

// everything but the tails: a workgroup per chunk, the chunk's window -- the 32 KB of text before it, which gz_tails_kernel has
// written -- in LDS, so that a marker costs an LDS look-up, not a one-byte load from global memory; eight symbols a thread and step
// (one 16-byte load, one 8-byte store; the addresses are two-byte / one-byte aligned only: gfx950 takes unaligned global accesses)
__global__ __launch_bounds__(256) void gz_resolve_kernel(const uint16_t *sym, uint8_t *out, const uint64_t *off, const uint64_t *len,
                                                          const uint32_t *chunk_file, const uint8_t *no_window, unsigned long long *first_nul)
{
    __shared__ uint8_t win[GZ_WIN];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    const uint32_t file = chunk_file[c];
    const bool whole = no_window[file] != 0;   // (a file of members only: no markers, no tails)
    const uint64_t o = off[c], L = len[c], body = whole ? L : L - (L < GZ_WIN ? L : GZ_WIN);
    if (body == 0) return;
    if (!whole)
        for (uint32_t i = tid * 16; i < GZ_WIN; i += 256 * 16) {
            uint4 v;
            __builtin_memcpy(&v, out + o - GZ_WIN + i, 16);
            *reinterpret_cast<uint4 *>(win + i) = v;
        }
    __syncthreads();
    for (uint64_t j0 = (uint64_t)tid * 8; j0 < body; j0 += 256 * 8) {
        const uint64_t p = o + j0;
        if (j0 + 8 <= body) {
            uint4 raw;
            __builtin_memcpy(&raw, sym + p, 16);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            uint64_t packed = 0;
            bool nul = false;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t s = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                const uint8_t b = s < 256 ? (uint8_t)s : win[(s - 256) & (GZ_WIN - 1)];   // (masked: a symbol of a failed chunk may be anything)
                packed |= (uint64_t)b << (8 * j);
                nul = nul || b == 0;
            }
            __builtin_memcpy(out + p, &packed, 8);
            if (nul)
                for (int j = 0; j < 8; j++)
                    if (((packed >> (8 * j)) & 0xff) == 0) atomicMin(first_nul + file, (unsigned long long)(p + j));
        } else {
            for (uint64_t j = j0; j < body; j++) {
                const uint32_t s = sym[o + j];
                const uint8_t b = s < 256 ? (uint8_t)s : win[(s - 256) & (GZ_WIN - 1)];
                out[o + j] = b;
                if (b == 0) atomicMin(first_nul + file, (unsigned long long)(o + j));
            }
        }
    }
}

// ---- step 6: the check sums ------------------------------------------------------------------------------------------
// zlib ends a member by comparing the CRC-32 of its text with the trailer (inflate.c: "incorrect data check"); so does this.
// A thread takes 4 KB of the text buffer: the CRC of every piece of a member inside it, moved to the member's end by a
// multiplication with x^(8 * bytes behind the piece) modulo the CRC polynomial (the algebra of zlib's crc32_combine: the CRC
// of a concatenation is the XOR of its pieces' CRCs so moved), XORed into the member's accumulator.
constexpr uint32_t GZ_CRC_POLY = 0xedb88320u;
constexpr int GZ_CRC_SEG = 4096;

__device__ __forceinline__ uint32_t gz_mulmod(uint32_t a, uint32_t b)   // a(x) * b(x) mod P, bit 31 = x^0
{
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? GZ_CRC_POLY : 0u);
    }
    return p;
}

__global__ __launch_bounds__(256) void gz_crc_kernel(const uint8_t *out, const uint64_t *m_begin, const uint64_t *m_len, int n_members, uint64_t first,
                                                      uint64_t total, uint32_t *acc)
{
    __shared__ uint32_t tab[4][256];
    __shared__ uint32_t x2n[32];   // x^(2^k) mod P
    for (int i = threadIdx.x; i < 256; i += 256) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? GZ_CRC_POLY : 0u);
        tab[0][i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 256) {
        uint32_t c = tab[0][i];
        for (int t = 1; t < 4; t++) {
            c = tab[0][c & 0xff] ^ (c >> 8);
            tab[t][i] = c;
        }
    }
    if (threadIdx.x == 0) {
        uint32_t p = 1u << 30;   // x^1
        x2n[0] = p;
        for (int k = 1; k < 32; k++) x2n[k] = p = gz_mulmod(p, p);
    }
    __syncthreads();
    const uint64_t p0 = first + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * GZ_CRC_SEG;
    const uint64_t p1 = p0 + GZ_CRC_SEG < total ? p0 + GZ_CRC_SEG : total;
    int lo = -1, hi = n_members;   // the first member that ends behind p0
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (m_begin[mid] + m_len[mid] > p0) hi = mid;
        else lo = mid;
    }
    // the first piece of a thread goes into its wave's sum where the whole wave works on one member (nearly always: a
    // member is megabytes, a wave's share 256 KB) -- a million atomics on one file's accumulator take their turns in the L2
    int my_m = -1;
    uint32_t my_v = 0;
    for (int m = hi; m < n_members && p0 < total; m++) {
        const uint64_t b = m_begin[m], e = b + m_len[m];
        if (b >= p1) break;
        const uint64_t q0 = b > p0 ? b : p0, q1 = e < p1 ? e : p1;
        if (q0 >= q1) continue;
        uint32_t c = 0xffffffffu;
        uint64_t q = q0;
        while (q < q1 && (q & 3)) c = tab[0][(c ^ out[q++]) & 0xff] ^ (c >> 8);
        while (q + 4 <= q1 && (q & 15)) {
            c ^= *reinterpret_cast<const uint32_t *>(out + q);
            c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
            q += 4;
        }
        for (; q + 128 <= q1; q += 128) {   // a whole line a lane: each line of the text is fetched once
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const uint4 *>(out + q + 16 * i);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c ^= w[j];
                    c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
                }
            }
        }
        for (; q + 16 <= q1; q += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(out + q);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c ^= w[j];
                c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
            }
        }
        for (; q + 4 <= q1; q += 4) {
            c ^= *reinterpret_cast<const uint32_t *>(out + q);
            c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24];
        }
        while (q < q1) c = tab[0][(c ^ out[q++]) & 0xff] ^ (c >> 8);
        c ^= 0xffffffffu;
        // behind the piece, inside the member: e - q1 bytes
        uint64_t nbytes = e - q1;
        uint32_t mul = 0x80000000u;   // x^0
        for (int k = 3; nbytes; nbytes >>= 1, k++)
            if (nbytes & 1) mul = gz_mulmod(x2n[k & 31], mul);
        const uint32_t v = gz_mulmod(mul, c);
        if (my_m < 0) {
            my_m = m;
            my_v = v;
        } else {
            atomicXor(acc + m, v);
        }
    }
    const int m0 = __builtin_amdgcn_readfirstlane(my_m);
    if (__ballot(my_m != m0 && my_m >= 0) == 0 && m0 >= 0) {
        uint32_t v = my_m == m0 ? my_v : 0u;
        for (int o = 32; o; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o, 64);
        if ((threadIdx.x & 63) == 0) atomicXor(acc + m0, v);
    } else if (my_m >= 0) {
        atomicXor(acc + my_m, my_v);
    }
}

// ---- guard bands (PSK_GZ_GUARD=1: the fuzz test's mode) ---------------------------------------------------------
// 64 KB of 0xA5 before and after the text, symbol and match buffers; counted after the last kernel: a decoder of untrusted
// bytes that writes one byte outside what the counting pass laid out is caught here (GPU AddressSanitizer is not available).
constexpr size_t GZ_GUARD = 65536;
constexpr int GZ_GUARD_BYTE = 0xA5;
struct GzGuardBands {
    const uint8_t *band[6];
};
__global__ __launch_bounds__(256) void gz_guard_check_kernel(GzGuardBands g, uint32_t *damaged)
{
    const uint8_t *b = g.band[blockIdx.y];
    uint32_t bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < GZ_GUARD; i += (size_t)gridDim.x * 256) bad += b[i] != (uint8_t)GZ_GUARD_BYTE;
    if (bad) atomicAdd(damaged, bad);
}

// ---- host side -----------------------------------------------------------------------------------------------
struct GzMemberHead {
    size_t deflate_at = 0;   // offset of the DEFLATE data in the file
    uint32_t bsize = 0;      // BGZF: length of the whole member (0: not a BGZF member)
};

// RFC 1952 2.3; false: no gzip member starts at `at`
bool gz_parse_member_header(const uint8_t *d, size_t n, size_t at, GzMemberHead *h)
{
    if (at + 18 > n || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8) return false;
    const uint8_t flg = d[at + 3];
    if (flg & 0xe0) return false;
    size_t p = at + 10;
    h->bsize = 0;
    if (flg & 4) {
        if (p + 2 > n) return false;
        const size_t xlen = d[p] | (d[p + 1] << 8);
        p += 2;
        if (p + xlen > n) return false;
        for (size_t q = p; q + 4 <= p + xlen;) {
            const size_t sl = d[q + 2] | (d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && sl == 2 && q + 6 <= p + xlen) h->bsize = (uint32_t)(d[q + 4] | (d[q + 5] << 8)) + 1;
            q += 4 + sl;
        }
        p += xlen;
    }
    if (flg & 8) {
        while (p < n && d[p]) p++;
        p++;
    }
    if (flg & 16) {
        while (p < n && d[p]) p++;
        p++;
    }
    if (flg & 2) p += 2;
    if (p + 8 > n) return false;
    h->deflate_at = p;
    return true;
}

uint32_t gz_le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct GzChunk {
    int file;
    uint64_t start_bit;   // in the device image (GZ_NONE until found)
    uint64_t seek_from, seek_to;   // bits: where gz_find_kernel looks (seek_from == seek_to: the start is known)
    bool true_start;
    // results of the counting pass
    uint64_t out_len = 0, n_rec = 0, end_bit = 0;
    int32_t link = GZ_ERROR;
    bool counted = false;
};

struct GzFile {
    const uint8_t *data;
    size_t size;
    uint64_t at;          // offset of the image in the device buffer
    bool bgzf = false;
    bool device_ok = true;
    std::vector<int> chain;      // its chunks in stream order
    uint64_t out_off = 0, out_len = 0;
    int nul_slot = -1;    // its entry of the writing pass's per-file tables
};

size_t gz_env(const char *name, size_t dflt)
{
    const char *s = std::getenv(name);
    return s && *s ? (size_t)std::strtoull(s, nullptr, 10) : dflt;
}

// zlib, every member of the file (what gzip.decompress and glistmaker's reader do); false: *err says why not
bool gz_host_inflate(const uint8_t *d, size_t n, std::vector<uint8_t> &out, std::string *err)
{
    out.clear();
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) {
        *err = "zlib: inflateInit2 failed";
        return false;
    }
    out.resize(std::max<size_t>(n * 4, (size_t)1 << 16));
    size_t ip = 0, have = 0;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        const uInt in_now = (uInt)std::min<size_t>(n - ip, (size_t)1 << 30), room = (uInt)std::min<size_t>(out.size() - have, (size_t)1 << 30);
        z.next_in = const_cast<Bytef *>(d + ip);
        z.avail_in = in_now;
        z.next_out = out.data() + have;
        z.avail_out = room;
        const int rc = inflate(&z, Z_NO_FLUSH);
        ip += in_now - z.avail_in;
        have += room - z.avail_out;
        if (rc == Z_STREAM_END) {
            while (ip < n && d[ip] == 0) ip++;   // padding; a further member?
            if (ip >= n) break;
            if (inflateReset(&z) != Z_OK) {
                inflateEnd(&z);
                *err = "zlib: inflateReset failed";
                return false;
            }
            continue;
        }
        if (rc == Z_OK || (rc == Z_BUF_ERROR && ip < n)) continue;
        *err = std::string("not a valid gzip file: ") + (rc == Z_BUF_ERROR ? "it ends inside a member" : (z.msg ? z.msg : "data error"));
        inflateEnd(&z);
        return false;
    }
    inflateEnd(&z);
    out.resize(have);
    return true;
}

}  // namespace

// Where the images of a group lie in its device buffer (16-byte aligned, at least 16 zero bytes behind each); returns its size.
uint64_t gz_image_layout(int n, const size_t *sizes, uint64_t *at)
{
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        at[i] = total;
        total += (sizes[i] + 16 + 15) & ~(size_t)15;
    }
    return total + 64;
}

// Whether a group is inflated on the device at all (else: zlib on host threads, nothing to upload).  The device route has a floor
// of ~40 ms -- its serial passes over one DEFLATE block take that long whatever the number of blocks -- and then runs at ~10 GB/s of
// compressed input; zlib inflates ~0.45 GB/s of text per thread, one file a thread (4 x the compressed bytes stand for the text).
// Measured (r05): 8 / 16 / 32 genomes of 5 Mbp as .fasta.gz: device 40 / 41 / 43 ms, eight host threads 12 / 37 / 77; four
// .fastq.gz files of 64 MB of text: 38 against 224 (four files keep four threads busy).  PSK_GZ_DEVICE_MIN_MB replaces the
// estimate by a threshold on the compressed megabytes.
bool gz_group_on_device(int n, const size_t *sizes, bool host_only, int host_threads)
{
    if (host_only || n <= 0) return false;
    size_t comp_bytes = 0;
    for (int i = 0; i < n; i++) comp_bytes += sizes[i];
    const char *fixed = std::getenv("PSK_GZ_DEVICE_MIN_MB");
    if (fixed && *fixed) return comp_bytes >= (gz_env("PSK_GZ_DEVICE_MIN_MB", 48) << 20);
    const int threads = std::max(1, std::min(n, std::min(host_threads < 1 ? 1 : host_threads, 32)));
    const double host_ms = 4.0 * (double)comp_bytes / (0.45e6 * threads), device_ms = 40.0 + (double)comp_bytes / 10e6;
    return device_ms < host_ms;
}

// Inflates n gzip images.  The text of file i is out_dev[res[i].off, + res[i].len) when res[i].on_device, else
// res[i].host (zlib on the host: the device route declined the file; *declined counts them).
int gz_inflate_group(psk_ctx *ctx, int n, const uint8_t *const *data, const size_t *sizes, DevBuf &comp_buf, DevBuf &sym_buf, DevBuf &rec_buf, DevBuf &out_buf,
                     DevBuf &tab_buf, std::vector<GzInflated> &res, double *device_ms, bool host_only, int host_threads, hipStream_t on_stream,
                     bool images_uploaded)
{
    res.assign((size_t)n, GzInflated());
    if (device_ms) *device_ms = 0.0;
    if (n <= 0) return PSK_OK;
    // Few blocks, few lanes: a DEFLATE block is decoded by ONE lane, three orders of magnitude slower than a host core decodes
    // it, so the device wins by numbers only (gz_group_on_device).  A group it would not win goes through zlib on
    // `host_threads` threads (what glistmaker does per file).
    size_t comp_bytes = 0;
    for (int i = 0; i < n; i++) comp_bytes += sizes[i];
    // zlib over the files `which` names, one file a thread of a pool of host_threads (the first error, in file order, is the call's)
    auto host_route = [&](const std::vector<int> &which) -> int {
        const int nw = (int)which.size();
        std::atomic<int> next(0), bad(0);
        std::vector<std::string> errs((size_t)nw);
        auto work = [&]() {
            for (;;) {
                const int j = next.fetch_add(1);
                if (j >= nw) return;
                const int i = which[(size_t)j];
                if (!gz_host_inflate(data[i], sizes[i], res[(size_t)i].host, &errs[(size_t)j])) {
                    bad = 1;
                    res[(size_t)i].error = errs[(size_t)j];   // (every file is tried: psk_gz_inflate reports per file)
                    res[(size_t)i].host.clear();
                }
                res[(size_t)i].len = res[(size_t)i].host.size();
            }
        };
        std::vector<std::thread> pool;
        const int nt = host_threads < 1 ? 1 : (host_threads > 32 ? 32 : host_threads);
        for (int t = 1; t < nt && t < nw; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (bad)
            for (int j = 0; j < nw; j++)
                if (!errs[(size_t)j].empty()) return psk_fail(ctx, PSK_EINVAL, "%s", errs[(size_t)j].c_str());
        return PSK_OK;
    };
    if (!gz_group_on_device(n, sizes, host_only, host_threads)) {
        std::vector<int> all((size_t)n);
        for (int i = 0; i < n; i++) all[(size_t)i] = i;
        PSK_TRY(host_route(all));
        if (std::getenv("PSK_TRACE"))
            std::fprintf(stderr, "[psk] gz inflate: %d files, %.1f MB compressed: zlib on at most %d host threads\n", n, comp_bytes / 1e6, host_threads < 1 ? 1 : host_threads);
        return PSK_OK;
    }
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = on_stream ? on_stream : ctx->stream;
    std::vector<GzFile> files((size_t)n);
    std::vector<uint64_t> image_at((size_t)n);
    const uint64_t comp_total = gz_image_layout(n, sizes, image_at.data());
    for (int i = 0; i < n; i++) {
        files[i].data = data[i];
        files[i].size = sizes[i];
        files[i].at = image_at[(size_t)i];
    }
    // ---- the chunks --------------------------------------------------------------------------------------------
    size_t deflate_bytes = 0;
    for (int i = 0; i < n; i++) deflate_bytes += sizes[i];
    const size_t want_lanes = gz_env("PSK_GZ_LANES", 65536);   // four waves on each of 256 CUs
    size_t chunk = gz_env("PSK_GZ_CHUNK", 0);
    if (!chunk) chunk = std::min<size_t>(std::max<size_t>(deflate_bytes / want_lanes, 16 << 10), 4 << 20);   // (16 KB: less than a block of most encoders -- two cuts in one block find the same start and one of the two lanes idles, but a group that cannot fill the part anyway is cut at every block: 1.15 GB of FASTQ text 69 -> 51 ms)
    chunk = (chunk + 3) & ~(size_t)3;
    std::vector<GzChunk> ch;
    std::vector<std::pair<int, int>> file_chunks((size_t)n);   // [first, last) of its regular chunks
    for (int i = 0; i < n; i++) {
        GzFile &f = files[i];
        GzMemberHead h;
        file_chunks[i] = {(int)ch.size(), (int)ch.size()};
        if (!gz_parse_member_header(f.data, f.size, 0, &h)) {
            f.device_ok = false;
            continue;
        }
        if (h.bsize) {
            // BGZF: every member says how long it is -- one chunk per member, nothing to search for
            f.bgzf = true;
            size_t at = 0;
            bool ok = true;
            while (at < f.size) {
                GzMemberHead m;
                if (!gz_parse_member_header(f.data, f.size, at, &m) || !m.bsize || at + m.bsize > f.size || m.deflate_at + 8 > at + m.bsize) {
                    ok = false;
                    break;
                }
                GzChunk c;
                c.file = i;
                c.start_bit = (f.at + m.deflate_at) * 8;
                c.seek_from = c.seek_to = 0;
                c.true_start = true;
                ch.push_back(c);
                at += m.bsize;
            }
            if (!ok) {
                ch.resize((size_t)file_chunks[i].first);
                f.bgzf = false;
                f.device_ok = false;
                continue;
            }
        } else {
            const size_t d0 = h.deflate_at, d1 = f.size - 8;
            for (size_t at = d0; at < d1 || at == d0; at += chunk) {
                GzChunk c;
                c.file = i;
                c.true_start = at == d0;
                c.start_bit = at == d0 ? (f.at + d0) * 8 : GZ_NONE;
                c.seek_from = at == d0 ? 0 : (f.at + at) * 8;
                c.seek_to = at == d0 ? 0 : (f.at + std::min(at + chunk, d1)) * 8;
                ch.push_back(c);
            }
        }
        file_chunks[i].second = (int)ch.size();
    }
    const auto t_begin = std::chrono::steady_clock::now();
    const bool trace = std::getenv("PSK_TRACE") != nullptr;
    auto t_last = t_begin;
    std::string phases;
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        std::snprintf(buf, sizeof buf, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        phases += buf;
        t_last = now;
    };
    // ---- the images ---------------------------------------------------------------------------------------------
    if (!images_uploaded) {   // (the counting calls upload each image as soon as it is read: gz_image_layout says where)
        PSK_TRY(dev_reserve(ctx, comp_buf, comp_total));
        PSK_HIP(ctx, hipMemsetAsync(comp_buf.p, 0, comp_total, st));
        for (int i = 0; i < n; i++)
            if (files[i].device_ok && sizes[i])
                PSK_HIP(ctx, hipMemcpyAsync(comp_buf.as<uint8_t>() + files[i].at, data[i], sizes[i], hipMemcpyHostToDevice, st));
    }
    const uint8_t *d_comp = comp_buf.as<uint8_t>();
    lap("upload");

    // device tables of one pass over `m` chunks, carved out of tab_buf
    auto carve = [&](size_t &off, size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    std::vector<int> todo;   // chunks of the coming counting pass
    for (size_t c = 0; c < ch.size(); c++) todo.push_back((int)c);
    std::vector<uint8_t> stage;
    int rounds = 0;
    const int max_rounds = (int)gz_env("PSK_GZ_ROUNDS", 24);
    // ---- step 1: the starts ---------------------------------------------------------------------------------------
    {
        std::vector<int> seek;
        for (size_t c = 0; c < ch.size(); c++)
            if (ch[c].seek_to > ch[c].seek_from) seek.push_back((int)c);
        if (!seek.empty()) {
            const size_t m = seek.size();
            size_t off = 0;
            const size_t o_from = carve(off, m * 8), o_to = carve(off, m * 8), o_end = carve(off, m * 8), o_found = carve(off, m * 8);
            PSK_TRY(dev_reserve(ctx, tab_buf, off));
            stage.assign(o_found, 0);
            for (size_t j = 0; j < m; j++) {
                const GzChunk &c = ch[(size_t)seek[j]];
                reinterpret_cast<uint64_t *>(stage.data() + o_from)[j] = c.seek_from;
                reinterpret_cast<uint64_t *>(stage.data() + o_to)[j] = c.seek_to;
                reinterpret_cast<uint64_t *>(stage.data() + o_end)[j] = files[(size_t)c.file].at + files[(size_t)c.file].size;
            }
            uint8_t *t = tab_buf.as<uint8_t>();
            PSK_HIP(ctx, hipMemcpyAsync(t, stage.data(), o_found, hipMemcpyHostToDevice, st));
            gz_find_kernel<<<dim3((unsigned)m), dim3(64), 0, st>>>(d_comp, reinterpret_cast<const uint64_t *>(t + o_from),
                                                                 reinterpret_cast<const uint64_t *>(t + o_to),
                                                                 reinterpret_cast<const uint64_t *>(t + o_end), (int)m,
                                                                 reinterpret_cast<uint64_t *>(t + o_found));
            PSK_HIP(ctx, hipGetLastError());
            std::vector<uint64_t> found(m);
            PSK_HIP(ctx, hipMemcpyAsync(found.data(), t + o_found, m * 8, hipMemcpyDeviceToHost, st));
            PSK_HIP(ctx, hipStreamSynchronize(st));
            for (size_t j = 0; j < m; j++) ch[(size_t)seek[j]].start_bit = found[j];
        }
    }
    lap("find");
    // the starts a block end may coincide with: per file, ascending
    std::vector<uint64_t> cand_bit;
    std::vector<int> cand_chunk;
    std::vector<std::pair<uint32_t, uint32_t>> file_cands((size_t)n);
    auto rebuild_cands = [&]() {
        cand_bit.clear();
        cand_chunk.clear();
        for (int i = 0; i < n; i++) {
            std::vector<std::pair<uint64_t, int>> v;
            for (size_t c = 0; c < ch.size(); c++)
                if (ch[c].file == i && ch[c].start_bit != GZ_NONE) v.push_back({ch[c].start_bit, (int)c});
            std::sort(v.begin(), v.end());
            file_cands[(size_t)i].first = (uint32_t)cand_bit.size();
            for (auto &e : v) {
                cand_bit.push_back(e.first);
                cand_chunk.push_back(e.second);
            }
            file_cands[(size_t)i].second = (uint32_t)cand_bit.size();
        }
    };
    // ---- steps 2 + 3: counting passes until every file's chain reaches its end ----------------------------------------
    // how far ONE lane may decode without meeting a found start.  A lane decodes ~1.4 MB of compressed input a second (65,536 of them:
    // 1.9 GB in 21 ms), so this is the bound on what a file with no dynamic block headers -- stored or fixed-code blocks only, or one
    // crafted to look so -- can cost the call before zlib gets it: 1 MiB = 0.75 s (r05: 8 MiB = 6 s, and again in every round of a
    // file of several such members).  Blocks of the usual encoders end far sooner (zlib: tens of KB; pigz 128 KB; libdeflate <= ~300 KB).
    const uint64_t max_span_bits = (uint64_t)std::max<size_t>(chunk * 8, gz_env("PSK_GZ_MAX_SPAN", 1 << 20)) * 8;
    std::vector<int> next_chunk((size_t)n, -1);   // per file: the chunk its chain follows next (-1: the chain has reached the end)
    for (int i = 0; i < n; i++) next_chunk[(size_t)i] = files[(size_t)i].device_ok && file_chunks[(size_t)i].second > file_chunks[(size_t)i].first ? file_chunks[(size_t)i].first : -1;
    while (!todo.empty()) {
        if (++rounds > max_rounds) {
            for (int c : todo) files[(size_t)ch[(size_t)c].file].device_ok = false;
            break;
        }
        rebuild_cands();
        const size_t m = todo.size();
        size_t off = 0;
        const size_t o_start = carve(off, m * 8), o_end = carve(off, m * 8), o_true = carve(off, m), o_cf = carve(off, m * 4), o_ct = carve(off, m * 4),
                     o_cand = carve(off, cand_bit.size() * 8 + 8);
        const size_t o_in_end = off;
        const size_t o_len = carve(off, m * 8), o_nrec = carve(off, m * 8), o_ebit = carve(off, m * 8), o_link = carve(off, m * 4),
                     o_long = carve(off, m * 640);
        PSK_TRY(dev_reserve(ctx, tab_buf, off));
        stage.assign(o_in_end, 0);
        for (size_t j = 0; j < m; j++) {
            const GzChunk &c = ch[(size_t)todo[j]];
            const GzFile &f = files[(size_t)c.file];
            reinterpret_cast<uint64_t *>(stage.data() + o_start)[j] = f.device_ok ? c.start_bit : GZ_NONE;
            reinterpret_cast<uint64_t *>(stage.data() + o_end)[j] = f.at + f.size;
            stage[o_true + j] = c.true_start ? 1 : 0;
            // its candidates: those of its file that start after it
            const auto &fc = file_cands[(size_t)c.file];
            const uint32_t from = (uint32_t)(std::upper_bound(cand_bit.begin() + fc.first, cand_bit.begin() + fc.second, c.start_bit) - cand_bit.begin());
            reinterpret_cast<uint32_t *>(stage.data() + o_cf)[j] = c.start_bit == GZ_NONE ? fc.second : from;
            reinterpret_cast<uint32_t *>(stage.data() + o_ct)[j] = fc.second;
        }
        if (!cand_bit.empty()) std::memcpy(stage.data() + o_cand, cand_bit.data(), cand_bit.size() * 8);
        uint8_t *t = tab_buf.as<uint8_t>();
        PSK_HIP(ctx, hipMemcpyAsync(t, stage.data(), o_in_end, hipMemcpyHostToDevice, st));
        GzDecodeArgs a;
        std::memset(&a, 0, sizeof a);
        a.comp = d_comp;
        a.start_bit = reinterpret_cast<const uint64_t *>(t + o_start);
        a.end_byte = reinterpret_cast<const uint64_t *>(t + o_end);
        a.true_start = t + o_true;
        a.cand_bit = reinterpret_cast<const uint64_t *>(t + o_cand);
        a.cand_from = reinterpret_cast<const uint32_t *>(t + o_cf);
        a.cand_to = reinterpret_cast<const uint32_t *>(t + o_ct);
        a.max_span_bits = max_span_bits;
        a.long_syms = reinterpret_cast<uint16_t *>(t + o_long);
        a.out_len = reinterpret_cast<uint64_t *>(t + o_len);
        a.n_rec = reinterpret_cast<uint64_t *>(t + o_nrec);
        a.end_bit = reinterpret_cast<uint64_t *>(t + o_ebit);
        a.link = reinterpret_cast<int32_t *>(t + o_link);
        a.n = (int)m;
        PSK_HIP(ctx, hipMemsetAsync(t + o_link, 0xfe, m * 4, st));   // (a chunk without a start keeps a negative link)
        gz_decode_kernel<false><<<dim3((unsigned)div_up((uint64_t)m, 64)), dim3(64), GZ_LDS_U16 * 2, st>>>(a);
        PSK_HIP(ctx, hipGetLastError());
        std::vector<uint64_t> r_len(m), r_ebit(m), r_nrec(m);
        std::vector<int32_t> r_link(m);
        PSK_HIP(ctx, hipMemcpyAsync(r_len.data(), t + o_len, m * 8, hipMemcpyDeviceToHost, st));
        PSK_HIP(ctx, hipMemcpyAsync(r_nrec.data(), t + o_nrec, m * 8, hipMemcpyDeviceToHost, st));
        PSK_HIP(ctx, hipMemcpyAsync(r_ebit.data(), t + o_ebit, m * 8, hipMemcpyDeviceToHost, st));
        PSK_HIP(ctx, hipMemcpyAsync(r_link.data(), t + o_link, m * 4, hipMemcpyDeviceToHost, st));
        PSK_HIP(ctx, hipStreamSynchronize(st));
        for (size_t j = 0; j < m; j++) {
            GzChunk &c = ch[(size_t)todo[j]];
            if (c.start_bit == GZ_NONE || !files[(size_t)c.file].device_ok) continue;
            c.counted = true;
            c.out_len = r_len[j];
            c.n_rec = r_nrec[j];
            c.end_bit = r_ebit[j];
            c.link = r_link[j] >= 0 ? cand_chunk[(size_t)r_link[j]] : r_link[j];
        }
        todo.clear();
        // the walk along the links
        for (int i = 0; i < n; i++) {
            GzFile &f = files[(size_t)i];
            while (f.device_ok && next_chunk[(size_t)i] >= 0) {
                const int cidx = next_chunk[(size_t)i];
                GzChunk &c = ch[(size_t)cidx];
                if (!c.counted) break;   // in `todo`: the next round
                f.chain.push_back(cidx);
                if (c.link >= 0) {
                    if (ch[(size_t)c.link].true_start) {
                        // a block that ends, without being the last of its member, where another member's data begin: no
                        // stream zlib accepts does that (a crafted BSIZE / member header does): zlib words the error
                        f.device_ok = false;
                        break;
                    }
                    next_chunk[(size_t)i] = c.link;
                    continue;
                }
                if (c.link != GZ_FINAL) {
                    f.device_ok = false;   // an error, or a block that ran on and on: zlib decides what it is
                    break;
                }
                // the member's trailer; another member behind it?
                const uint64_t trailer = (c.end_bit + 7) / 8 - f.at;
                if (trailer + 8 > f.size) {
                    f.device_ok = false;
                    break;
                }
                size_t next = (size_t)trailer + 8;
                {
                    // ISIZE of the member that just ended: the bytes since its true start
                    uint64_t member = 0;
                    for (size_t q = f.chain.size(); q-- > 0;) {
                        member += ch[(size_t)f.chain[q]].out_len;
                        if (ch[(size_t)f.chain[q]].true_start) break;
                    }
                    if ((uint32_t)member != gz_le32(f.data + trailer + 4)) {
                        f.device_ok = false;
                        break;
                    }
                }
                while (next < f.size && f.data[next] == 0) next++;   // padding (gzip.decompress skips it as well)
                if (next >= f.size) {
                    next_chunk[(size_t)i] = -1;
                    break;
                }
                if (f.bgzf) {
                    // the next member is the next chunk of the file
                    next_chunk[(size_t)i] = cidx + 1 < file_chunks[(size_t)i].second ? cidx + 1 : -1;
                    if (next_chunk[(size_t)i] < 0) f.device_ok = false;
                    continue;
                }
                GzMemberHead h;
                if (!gz_parse_member_header(f.data, f.size, next, &h)) {
                    f.device_ok = false;
                    break;
                }
                // a member in the middle of the file: it starts a chain of its own (a chunk whose found start this is
                // becomes its first link; else one more chunk, counted in the next round)
                const uint64_t sbit = (f.at + h.deflate_at) * 8;
                int have = -1;
                for (int q = file_chunks[(size_t)i].first; q < (int)ch.size(); q++)
                    if (ch[(size_t)q].file == i && ch[(size_t)q].start_bit == sbit) have = q;
                if (have >= 0) {
                    ch[(size_t)have].true_start = true;   // (what it counted stays right: a valid stream has no match reaching back here)
                    next_chunk[(size_t)i] = have;
                    continue;
                }
                GzChunk extra;
                extra.file = i;
                extra.start_bit = sbit;
                extra.seek_from = extra.seek_to = 0;
                extra.true_start = true;
                ch.push_back(extra);
                todo.push_back((int)ch.size() - 1);
                next_chunk[(size_t)i] = (int)ch.size() - 1;
                break;
            }
        }
    }
    lap("count");
    // ---- the layout of the text -----------------------------------------------------------------------------------
    uint64_t total = GZ_WIN;   // (room before the first file: a marker of a corrupt stream reads inside the buffer)
    std::vector<int> order;    // the chunks of the writing pass, file by file, in stream order
    std::vector<uint32_t> file_first;
    std::vector<uint64_t> c_off, c_rec, m_begin, m_len;   // (m_*: the members, in the order of the text buffer)
    std::vector<uint32_t> m_crc;
    std::vector<int> m_file;
    uint64_t total_rec = 0;
    // a file's chain is members one after the other -- first chunk a member's start, none in the middle of a member (a chunk
    // that was linked to BEFORE a later walk found a member starting there), the last chunk the end of a member: anything else
    // is declined here, so that every member below has a beginning AND a length and the check sums are never skipped
    for (int i = 0; i < n; i++) {
        GzFile &f = files[(size_t)i];
        if (!f.device_ok) continue;
        bool open = false, ok = !f.chain.empty();
        for (int c : f.chain) {
            const GzChunk &k = ch[(size_t)c];
            if (k.true_start == open) ok = false;   // a start inside a member, or a member that does not begin with one
            open = k.link != GZ_FINAL;
        }
        if (!ok || open) f.device_ok = false;
    }
    for (int i = 0; i < n; i++) {
        GzFile &f = files[(size_t)i];
        if (!f.device_ok) continue;
        total = (total + 63) & ~63ull;
        f.out_off = total;
        f.nul_slot = (int)file_first.size();
        file_first.push_back((uint32_t)order.size());
        for (int c : f.chain) {
            const GzChunk &k = ch[(size_t)c];
            order.push_back(c);
            c_off.push_back(total);
            if (k.true_start) m_begin.push_back(total);
            total += k.out_len;
            c_rec.push_back(total_rec);
            total_rec += k.n_rec;
            if (k.link == GZ_FINAL) {   // the member's trailer: CRC-32, ISIZE
                m_len.push_back(total - m_begin.back());
                m_crc.push_back(gz_le32(f.data + ((k.end_bit + 7) / 8 - f.at)));
                m_file.push_back(i);
            }
        }
        f.out_len = total - f.out_off;
    }
    file_first.push_back((uint32_t)order.size());
    total = (total + 63) & ~63ull;
    const size_t m = order.size();
    std::vector<unsigned long long> nul_at;   // per file of the writing pass: where its text has its first NUL (in out_buf), ~0: nowhere
    const size_t guard = std::getenv("PSK_GZ_GUARD") ? GZ_GUARD : 0;
    uint32_t guard_damage = 0;
    if (m) {
        PSK_TRY(dev_reserve(ctx, sym_buf, total * 2 + 64 + 2 * guard));
        PSK_TRY(dev_reserve(ctx, out_buf, total + 64 + 2 * guard));
        PSK_TRY(dev_reserve(ctx, rec_buf, total_rec * 8 + 64 + 2 * guard));
        // (the buffers proper start behind the leading band; the 64 bytes of slack the wide loads may touch stay in front of the trailing one)
        uint16_t *const d_sym = reinterpret_cast<uint16_t *>(sym_buf.as<uint8_t>() + guard);
        uint8_t *const d_text = out_buf.as<uint8_t>() + guard;
        uint2 *const d_rec = reinterpret_cast<uint2 *>(rec_buf.as<uint8_t>() + guard);
        GzGuardBands bands;
        if (guard) {
            uint8_t *b[6] = {sym_buf.as<uint8_t>(), sym_buf.as<uint8_t>() + guard + total * 2 + 64, out_buf.as<uint8_t>(), out_buf.as<uint8_t>() + guard + total + 64,
                             rec_buf.as<uint8_t>(), rec_buf.as<uint8_t>() + guard + total_rec * 8 + 64};
            for (int q = 0; q < 6; q++) {
                bands.band[q] = b[q];
                PSK_HIP(ctx, hipMemsetAsync(b[q], GZ_GUARD_BYTE, guard, st));
            }
        }
        lap("buffers");
        size_t off = 0;
        const size_t o_start = carve(off, m * 8), o_end = carve(off, m * 8), o_true = carve(off, m), o_stop = carve(off, m * 8), o_off = carve(off, m * 8),
                     o_want = carve(off, m * 8), o_roff = carve(off, m * 8), o_wrec = carve(off, m * 8), o_ff = carve(off, file_first.size() * 4),
                     o_cfile = carve(off, m * 4), o_nul = carve(off, file_first.size() * 8), o_nowin = carve(off, file_first.size());
        const size_t o_in_end = off;
        const size_t o_len = carve(off, m * 8), o_nrec = carve(off, m * 8), o_ebit = carve(off, m * 8), o_link = carve(off, m * 4),
                     o_long = carve(off, m * 640);
        const size_t o_members = carve(off, 2 * ((m_begin.size() * 8 + 255) & ~(size_t)255) + m_begin.size() * 4 + 256);
        const size_t o_guard = carve(off, 256);
        PSK_TRY(dev_reserve(ctx, tab_buf, off));
        stage.assign(o_in_end, 0);
        for (size_t j = 0; j < m; j++) {
            const GzChunk &c = ch[(size_t)order[j]];
            const GzFile &f = files[(size_t)c.file];
            reinterpret_cast<uint64_t *>(stage.data() + o_start)[j] = c.start_bit;
            reinterpret_cast<uint64_t *>(stage.data() + o_end)[j] = f.at + f.size;
            stage[o_true + j] = c.true_start ? 1 : 0;
            reinterpret_cast<uint64_t *>(stage.data() + o_stop)[j] = c.link == GZ_FINAL ? GZ_NONE : c.end_bit;
            reinterpret_cast<uint64_t *>(stage.data() + o_off)[j] = c_off[j];
            reinterpret_cast<uint64_t *>(stage.data() + o_want)[j] = c.out_len;
            reinterpret_cast<uint64_t *>(stage.data() + o_roff)[j] = c_rec[j];
            reinterpret_cast<uint64_t *>(stage.data() + o_wrec)[j] = c.n_rec;
        }
        std::memcpy(stage.data() + o_ff, file_first.data(), file_first.size() * 4);
        for (size_t fi = 0; fi + 1 < file_first.size(); fi++)
            for (uint32_t j = file_first[fi]; j < file_first[fi + 1]; j++) reinterpret_cast<uint32_t *>(stage.data() + o_cfile)[j] = (uint32_t)fi;
        std::memset(stage.data() + o_nul, 0xff, file_first.size() * 8);
        for (size_t fi = 0; fi + 1 < file_first.size(); fi++) {
            bool members_only = true;
            for (uint32_t j = file_first[fi]; j < file_first[fi + 1]; j++) members_only = members_only && ch[(size_t)order[j]].true_start;
            stage[o_nowin + fi] = members_only ? 1 : 0;
        }
        uint8_t *t = tab_buf.as<uint8_t>();
        PSK_HIP(ctx, hipMemcpyAsync(t, stage.data(), o_in_end, hipMemcpyHostToDevice, st));
        GzDecodeArgs a;
        std::memset(&a, 0, sizeof a);
        a.comp = d_comp;
        a.start_bit = reinterpret_cast<const uint64_t *>(t + o_start);
        a.end_byte = reinterpret_cast<const uint64_t *>(t + o_end);
        a.true_start = t + o_true;
        a.stop_bit = reinterpret_cast<const uint64_t *>(t + o_stop);
        a.out_off = reinterpret_cast<const uint64_t *>(t + o_off);
        a.want_len = reinterpret_cast<const uint64_t *>(t + o_want);
        a.rec_off = reinterpret_cast<const uint64_t *>(t + o_roff);
        a.want_rec = reinterpret_cast<const uint64_t *>(t + o_wrec);
        a.sym = d_sym;
        a.rec = d_rec;
        a.long_syms = reinterpret_cast<uint16_t *>(t + o_long);
        a.out_len = reinterpret_cast<uint64_t *>(t + o_len);
        a.n_rec = reinterpret_cast<uint64_t *>(t + o_nrec);
        a.end_bit = reinterpret_cast<uint64_t *>(t + o_ebit);
        a.link = reinterpret_cast<int32_t *>(t + o_link);
        a.n = (int)m;
        gz_decode_kernel<true><<<dim3((unsigned)div_up((uint64_t)m, 64)), dim3(64), GZ_LDS_U16 * 2, st>>>(a);
        PSK_HIP(ctx, hipGetLastError());
        const uint64_t *d_off = reinterpret_cast<const uint64_t *>(t + o_off), *d_len = reinterpret_cast<const uint64_t *>(t + o_want);
        unsigned long long *d_stats = nullptr;
        if (std::getenv("PSK_GZ_STATS")) {
            PSK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d_stats), 64));
            PSK_HIP(ctx, hipMemsetAsync(d_stats, 0, 64, st));
        }
        gz_copy_kernel<<<dim3((unsigned)div_up((uint64_t)m, 4)), dim3(256), 0, st>>>(d_sym, d_rec, a.rec_off, a.want_rec, a.n_rec, a.link,
                                                                                   a.want_len, d_off, (int)m, d_stats);
        PSK_HIP(ctx, hipGetLastError());
        if (d_stats) {
            unsigned long long hs[8];
            PSK_HIP(ctx, hipStreamSynchronize(st));
            PSK_HIP(ctx, hipMemcpy(hs, d_stats, 64, hipMemcpyDeviceToHost));
            (void)hipFree(d_stats);
            std::fprintf(stderr, "[psk] gz copy: %llu matches in %llu groups of 64, %llu with a source inside the group, %llu redirection hops, %llu matches left waiting, "
                                 "%llu rounds\n",
                         (unsigned long long)total_rec, hs[0], hs[2], hs[3], hs[5], hs[1]);
        }
        unsigned long long *d_nul = reinterpret_cast<unsigned long long *>(t + o_nul);
        gz_tails_kernel<<<dim3((unsigned)(file_first.size() - 1)), dim3(1024), 0, st>>>(d_sym, d_text, d_off, d_len,
                                                                                     reinterpret_cast<const uint32_t *>(t + o_ff), t + o_nowin, d_nul);
        PSK_HIP(ctx, hipGetLastError());
        gz_resolve_kernel<<<dim3((unsigned)m), dim3(256), 0, st>>>(d_sym, d_text, d_off, d_len,
                                                                   reinterpret_cast<const uint32_t *>(t + o_cfile), t + o_nowin, d_nul);
        PSK_HIP(ctx, hipGetLastError());
        nul_at.resize(file_first.size());
        PSK_HIP(ctx, hipMemcpyAsync(nul_at.data(), d_nul, file_first.size() * 8, hipMemcpyDeviceToHost, st));
        // the members' check sums (the tables go behind the writing pass's: tab_buf is grown before the pass, not here)
        std::vector<uint32_t> crc_got(m_begin.size(), 0);
        if (m_begin.size() != m_len.size() || m_len.size() != m_crc.size())
            return psk_fail(ctx, PSK_ESTATE, "gz inflate: %zu member starts, %zu member ends (internal error)", m_begin.size(), m_len.size());
        const bool check_crc = !m_begin.empty() && !std::getenv("PSK_GZ_NO_CRC");
        if (check_crc) {
            const size_t nm = m_begin.size();
            uint8_t *mt = tab_buf.as<uint8_t>() + o_members;
            PSK_HIP(ctx, hipMemcpyAsync(mt, m_begin.data(), nm * 8, hipMemcpyHostToDevice, st));
            PSK_HIP(ctx, hipMemcpyAsync(mt + ((nm * 8 + 255) & ~(size_t)255), m_len.data(), nm * 8, hipMemcpyHostToDevice, st));
            uint32_t *d_acc = reinterpret_cast<uint32_t *>(mt + 2 * ((nm * 8 + 255) & ~(size_t)255));
            PSK_HIP(ctx, hipMemsetAsync(d_acc, 0, nm * 4, st));
            const uint64_t span = total - GZ_WIN;   // (nothing but empty texts: no launch; the check sums of empty members are 0)
            if (span)
                gz_crc_kernel<<<dim3((unsigned)div_up(span, (uint64_t)256 * GZ_CRC_SEG)), dim3(256), 0, st>>>(
                d_text, reinterpret_cast<const uint64_t *>(mt), reinterpret_cast<const uint64_t *>(mt + ((nm * 8 + 255) & ~(size_t)255)), (int)nm,
                (uint64_t)GZ_WIN, total, d_acc);
            PSK_HIP(ctx, hipGetLastError());
            PSK_HIP(ctx, hipMemcpyAsync(crc_got.data(), d_acc, nm * 4, hipMemcpyDeviceToHost, st));
        }
        std::vector<int32_t> r_link(m);
        PSK_HIP(ctx, hipMemcpyAsync(r_link.data(), t + o_link, m * 4, hipMemcpyDeviceToHost, st));
        if (guard) {
            uint32_t *d_damaged = reinterpret_cast<uint32_t *>(t + o_guard);
            PSK_HIP(ctx, hipMemsetAsync(d_damaged, 0, 4, st));
            gz_guard_check_kernel<<<dim3(16, 6), dim3(256), 0, st>>>(bands, d_damaged);
            PSK_HIP(ctx, hipGetLastError());
            PSK_HIP(ctx, hipMemcpyAsync(&guard_damage, d_damaged, 4, hipMemcpyDeviceToHost, st));
        }
        PSK_HIP(ctx, hipStreamSynchronize(st));
        lap("write + matches + markers + crc");
        if (guard_damage)
            return psk_fail(ctx, PSK_ESTATE, "gz inflate: %u bytes of the guard bands around the text / symbol / match buffers were overwritten", guard_damage);
        for (size_t j = 0; j < m; j++)
            if (r_link[j] == GZ_ERROR)   // the second decode disagrees with the first: nothing of this file is trusted
                files[(size_t)ch[(size_t)order[j]].file].device_ok = false;
        if (check_crc)
            for (size_t q = 0; q < m_crc.size(); q++)
                if (crc_got[q] != m_crc[q]) files[(size_t)m_file[q]].device_ok = false;   // zlib will say "incorrect data check"
    }
    if (device_ms) *device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    std::vector<int> declined_files;
    for (int i = 0; i < n; i++) {
        const GzFile &f = files[(size_t)i];
        GzInflated &r = res[(size_t)i];
        r.on_device = f.device_ok;
        r.bgzf = f.bgzf;
        r.chunks = (int)f.chain.size();
        if (f.device_ok) {
            r.off = f.out_off + guard;   // (relative to the buffer the caller holds: the leading band is in front)
            r.len = f.out_len;
            r.first_nul = r.len;
            if (f.nul_slot >= 0 && (size_t)f.nul_slot < nul_at.size() && nul_at[(size_t)f.nul_slot] != ~0ull) r.first_nul = nul_at[(size_t)f.nul_slot] - f.out_off;
        } else {
            declined_files.push_back(i);
        }
    }
    // what the device declined goes through zlib on the call's host threads, like a group the device never saw (a
    // run of read sets from an encoder the device route declines -- fixed-Huffman or stored blocks only -- paid one thread)
    const int declined = (int)declined_files.size();
    if (declined) PSK_TRY(host_route(declined_files));
    if (trace)
        std::fprintf(stderr, "[psk] gz inflate: %d files, %zu chunks of %zu KB, %d counting round(s), %d declined (zlib on the host); ms:%s\n", n, ch.size(),
                     chunk >> 10, rounds, declined, phases.c_str());
    return PSK_OK;
}

void gz_release_device(psk_ctx *ctx)
{
    const auto t0 = std::chrono::steady_clock::now();
    size_t dev_bytes = 0;
    for (DevBuf *b : {&ctx->gz_comp[0], &ctx->gz_comp[1], &ctx->gz_out[0], &ctx->gz_out[1], &ctx->gz_sym, &ctx->gz_rec, &ctx->gz_tab}) {
        dev_bytes += b->cap;
        dev_release(*b);
    }
    if (dev_bytes && std::getenv("PSK_TRACE"))
        std::fprintf(stderr, "[psk] gz release: %.1f GB of device buffers in %.1f ms\n", dev_bytes / 1e9,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

void gz_release_host(psk_ctx *ctx, bool wait)
{
    const bool trace = std::getenv("PSK_TRACE") != nullptr;
    {
        const auto m0 = std::chrono::steady_clock::now();
        size_t bytes = 0;
        for (auto &set : ctx->gz_maps) {
            for (auto &m : set) {
                munmap(m.first, m.second);
                bytes += m.second;
            }
            set.clear();
        }
        if (trace && bytes)
            std::fprintf(stderr, "[psk] gz release: %.1f GB of mapped .gz files unmapped in %.1f ms\n", bytes / 1e9,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m0).count());
    }
    if (ctx->gz_reaper.joinable()) ctx->gz_reaper.join();
    uint8_t *gone[2] = {ctx->gz_host[0], ctx->gz_host[1]};
    size_t host_bytes = 0;
    for (int q = 0; q < 2; q++) {
        host_bytes += ctx->gz_host_cap[q];
        ctx->gz_host[q] = nullptr;
        ctx->gz_host_cap[q] = 0;
    }
    if (!gone[0] && !gone[1]) return;
    ctx->gz_reaper = std::thread([gone, trace, host_bytes] {
        const auto h0 = std::chrono::steady_clock::now();
        std::free(gone[0]);
        std::free(gone[1]);
        if (trace)
            std::fprintf(stderr, "[psk] gz release: %.1f GB of host buffers given back in %.1f ms (helper thread)\n", host_bytes / 1e9,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count());
    });
    if (wait) ctx->gz_reaper.join();
}

void gz_release(psk_ctx *ctx)
{
    gz_release_device(ctx);
    gz_release_host(ctx, true);
}

// ---- C-ABI: the inflate on its own (tests, measurements) ------------------------------------------------------------
extern "C" int psk_gz_inflate(psk_ctx *ctx, int n, const uint8_t *const *data, const size_t *sizes, uint8_t *const *out, const size_t *out_cap,
                              uint64_t *out_len, int32_t *route, double *device_ms)
{
    if (!ctx) return PSK_EINVAL;
    if (n < 0 || (n && (!data || !sizes || !out_len))) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    std::vector<GzInflated> res;
    // (the context's buffers, kept for the next call: psk_build_presence / psk_free give them back)
    DevBuf &outb = ctx->gz_out[0];
    int rc = gz_inflate_group(ctx, n, data, sizes, ctx->gz_comp[0], ctx->gz_sym, ctx->gz_rec, ctx->gz_out[0], ctx->gz_tab, res, device_ms);
    // a file zlib refuses fails the call (PSK_EINVAL, zlib's words for the first such file) -- but every file has been tried: with
    // `route` given, route[i] = -1 marks the refused ones and the others' texts are delivered all the same
    bool per_file = false;
    if (rc == PSK_EINVAL && route && res.size() == (size_t)n)
        for (const GzInflated &r : res) per_file = per_file || !r.error.empty();
    if (rc == PSK_OK || per_file) {
        const int rc_files = rc;
        rc = PSK_OK;
        for (int i = 0; i < n && rc == PSK_OK; i++) {
            const GzInflated &r = res[(size_t)i];
            out_len[i] = r.len;
            if (route) route[i] = !r.error.empty() ? -1 : r.on_device ? (r.bgzf ? 2 : 1) : 0;
            if (!out || !out[i] || !r.error.empty()) continue;
            if (!out_cap || out_cap[i] < r.len) {
                rc = psk_fail(ctx, PSK_ERANGE, "file %d inflates to %llu bytes, the buffer holds %llu", i, (unsigned long long)r.len,
                              (unsigned long long)(out_cap ? out_cap[i] : 0));
                break;
            }
            if (r.on_device) {
                if (r.len && hipMemcpy(out[i], outb.as<uint8_t>() + r.off, r.len, hipMemcpyDeviceToHost) != hipSuccess)
                    rc = psk_fail(ctx, PSK_EHIP, "copying the inflated text failed");
            } else if (r.len) {
                std::memcpy(out[i], r.host.data(), r.len);
            }
        }
        if (rc == PSK_OK) rc = rc_files;   // (the message of the first refused file is still the context's last error)
    }
    return rc;
}
