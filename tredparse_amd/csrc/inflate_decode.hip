// inflate_decode.hip -- raw DEFLATE (RFC 1951) decoding of many BGZF blocks at once on gfx950.
//
// The read-selection front end of the path (/root/reference/tredparse/bam_parser.py:184-257, 316-369: pysam fetch /
// pileup, i.e. htslib's bgzf_read -> zlib inflate) spends two thirds of its host time inflating BGZF blocks: 35 MB
// per 30x sample in ~550 independent blocks of at most 64 KiB, on a box whose 16 host cores -- not its GPU -- bound the
// end-to-end rate.  Round 3 decoded one block per LANE (Huffman tables per lane in LDS: 127 KB per wavefront, one
// wavefront per CU, every memory instruction touching 64 cache lines, 21 ms per wavefront).  This is the round-4
// decoder: ONE WAVEFRONT = ONE BLOCK, built around what is serial in DEFLATE and what is not.
//
//   * What is serial is only WHERE the next symbol starts.  What a symbol IS, given its start, is not: so every lane
//     decodes the complete symbol that would start at ITS bit offset of a 64-bit window of the stream -- literal /
//     length code through a 9-bit root table in LDS and, for longer codes, a second-level table behind a link (four
//     look-ups per lane at most, no branches), the length's extra bits, the distance code (7-bit root) and its extra
//     bits, all from the lane's own 57-bit view of the stream (15 + 5 + 15 + 13 = 48 bits at most) -- and packs (bits
//     consumed, kind, length or literal, distance) into one dword.  Most of the 64 answers are for offsets no symbol
//     starts at: they are what lets the serial part be as short as it is.
//   * The serial chain is then a walk over lanes on the scalar unit: v_readlane the dword at the current offset, set
//     the offset's bit in a 64-bit mask of symbol starts, add the symbol's bit count to the offset -- no table look-up,
//     no memory access, four instructions and a branch per symbol; the windows' look-ups do not depend on it.  The
//     lanes whose bit is set then append their dword to a queue in LDS (rank = prefix popcount of the mask).
//   * The queue is executed 64 symbols at a time by the whole wavefront: an inclusive DPP scan of the output
//     lengths gives every symbol its destination; literals are one byte store; matches of at most 16 bytes whose
//     source lies before the batch copy themselves (one unaligned 16-byte load, two overlapping exact-length stores);
//     long matches and the few whose source reaches into the batch are copied one after the other by all 64 lanes,
//     64 bytes per step (period handling for distances below 64).  Destinations of a batch are consecutive, so the
//     stores of a batch fall into a handful of cache lines.
//   * Tables are built by the wavefront together: code-length histogram by LDS atomics, every symbol its own canonical
//     code (rank within its length by ballot + prefix popcount) and the entries that decode to it; the second-level
//     tables of both alphabets share 640 entries (the worst cases of complete codes need 340 + 272; a block that asked
//     for more would get status -3 and be inflated by the host like any block the decoder does not vouch for).
//   * LDS: 5.6 KB per wavefront and 72 VGPRs: 28 wavefronts per CU instead of one -- a wavefront is a chain of
//     short dependent steps, what fills the CU is how many are resident; no per-block workspace in global memory.
// Nothing here knows BAM: the C ABI (include/tredgpu.h, tredgpu_inflate_*) takes payload offsets and sizes and
// returns bytes plus a status per block; gzip framing and ISIZE stay with the host library (libtredbam).
#include <hip/hip_runtime.h>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <cstdio>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "inflater_internal.h"

using namespace tredgpu_front;

namespace {

constexpr int MAXBITS = 15, MAXL = 288, MAXD = 32;
#ifndef ROOTL_BITS
#define ROOTL_BITS 9
#define ROOTD_BITS 7
#endif
constexpr int ROOTL = ROOTL_BITS, ROOTD = ROOTD_BITS;
constexpr int SUB_CAP = 640;       // entries of second-level tables, both alphabets together: complete codes of 286 / 30 symbols
                                   // and at most 15 bits need 340 behind a 9-bit root (zlib's ENOUGH_LENS 852 - 512) and 272
                                   // behind a 7-bit one at the worst (exhaustive hill climbing over code length sets)

// one wavefront's tables (one block in flight per wavefront).  5.6 KB: 28 wavefronts per CU -- the decoder is a chain
// of short dependent steps per wavefront, and what fills the CU is how many of them are resident
struct WaveLds {
    uint32_t rootL[1 << ROOTL];   // direct tables on the next ROOTL / ROOTD bits of the stream (entries: see entry_L / entry_D);
    uint32_t rootD[1 << ROOTD];   // a code longer than that: a LINK to its second-level table in sub[]
    uint32_t sub[SUB_CAP];        // second-level tables of both alphabets, on the bits behind the root's
    union {
        uint32_t queue[2 * LANES];    // decoded symbols in stream order, waiting to be executed 64 at a time
        struct {                      // what only the block header needs (the queue is empty then)
            uint32_t cnt[16];             // codes per length of the alphabet under construction
            uint8_t lens[MAXL + MAXD];    // code lengths as the header gives them
            uint8_t clsym[20];            // the code-length code: sorted symbols and its 7-bit direct table (symbol | length << 5)
            uint8_t clfast[128];
        } hdr;
    };
};

// packed symbol: [31:25] bits consumed, [24:23] kind, [22:8] distance - 1, [7:0] literal or match length - 3.  The
// end-of-block code and "no such code" say 64 bits consumed -- the walk over a window stops at them by itself -- and
// the end-of-block code keeps its real length in the low byte.
enum : uint32_t { K_LIT = 0, K_MATCH = 1, K_END = 2, K_BAD = 3 };
constexpr int P_BITS = 25, P_KIND = 23, P_DIST = 8;
constexpr uint32_t BAD_SYMBOL = 64u << P_BITS | K_BAD << P_KIND;

// base value and extra bits of length code c (0..28) / distance code d (0..29), RFC 1951 3.2.5, in closed form
__device__ __forceinline__ void len_code(int c, int& base, int& extra) {
    extra = c < 8 || c == 28 ? 0 : (c >> 2) - 1;
    base = c < 8 ? 3 + c : (c == 28 ? 258 : 3 + ((4 + (c & 3)) << extra));
}
__device__ __forceinline__ void dist_code(int d, int& base, int& extra) {
    extra = d < 4 ? 0 : (d >> 1) - 1;
    base = d < 4 ? 1 + d : 1 + ((2 + (d & 1)) << extra);
}
__constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// any-alignment accesses (global memory takes them on gfx950)
typedef uint64_t __attribute__((aligned(1))) U64;
typedef uint32_t __attribute__((aligned(1))) U32;
typedef uint16_t __attribute__((aligned(1))) U16;

__device__ __forceinline__ int lanes_below(uint64_t m) {   // set bits of m in lanes below this one
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// The block header is read by the whole wavefront in step: every value here is wave-uniform (scalar registers, the
// payload through the scalar cache).
struct UBits {
    const uint32_t* p;
    int nwords, idx;       // dwords that hold payload, next dword to fetch
    uint64_t buf;
    int cnt;
    __device__ __forceinline__ uint32_t word(int i) const { return i < nwords ? p[i] : 0u; }   // past the payload: zeros
    __device__ __forceinline__ void start(const uint32_t* at, int words, int bit) {
        p = at; nwords = words; idx = (bit >> 5) + 1;
        buf = (uint64_t)(word(idx - 1) >> (bit & 31));
        cnt = 32 - (bit & 31);
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) { buf |= (uint64_t)word(idx) << cnt; cnt += 32; ++idx; }
    }
    __device__ __forceinline__ uint32_t peek(int n) { refill(); return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void skip(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(int n) { const uint32_t v = peek(n); skip(n); return v; }   // n <= 16
    __device__ __forceinline__ int pos() const { return idx * 32 - cnt; }
};

// Table entries (one dword) carry everything a lane needs, so that the window's look-ups are a handful of
// instructions per lane:
//   literal / length table:  literal      -> the packed symbol itself
//                            end of block -> the packed symbol itself
//                            length code  -> [31:25] code length + extra bits, [24:23] K_MATCH, [22:20] extra bits,
//                                            [19:16] code length, [7:0] base length - 3
//   distance table:          [31:25] code length + extra bits, [23:20] extra bits, [19:16] code length, [14:0] base - 1
//   both:                    LINK | bits << 16 | first: the code is longer than the root table's bits -- its entry is
//                            sub[first + the next `bits` bits of the stream] (round 4 decoded such codes canonically,
//                            only where a symbol really started with one: 4-5 % of the symbols, a third of the kernel's
//                            vector instructions);  BAD_SYMBOL = a symbol no stream may use, a code that is none.
//                            No valid entry has its top bit set: the walk stops at BAD by itself; LINK has the top two.
constexpr uint32_t LINK = 0xC0000000u;

__device__ __forceinline__ uint32_t entry_L(int sym, int clen) {
    if (sym < 256) return (uint32_t)clen << P_BITS | K_LIT << P_KIND | (uint32_t)sym;
    if (sym == 256) return 64u << P_BITS | K_END << P_KIND | (uint32_t)clen;
    const int c = sym - 257;
    if (c >= 29) return BAD_SYMBOL;
    int base, extra;
    len_code(c, base, extra);
    return (uint32_t)(clen + extra) << P_BITS | K_MATCH << P_KIND | (uint32_t)extra << 20 | (uint32_t)clen << 16 | (uint32_t)(base - 3);
}
__device__ __forceinline__ uint32_t entry_D(int sym, int clen) {
    if (sym >= 30) return BAD_SYMBOL;
    int base, extra;
    dist_code(sym, base, extra);
    return (uint32_t)(clen + extra) << P_BITS | (uint32_t)extra << 20 | (uint32_t)clen << 16 | (uint32_t)(base - 1);
}

// Root table and second-level tables of a canonical code from n code lengths in LDS, by the whole wavefront: every
// symbol works out its own code (first code of its length + its rank among the symbols of that length: a ballot and a
// prefix popcount) and writes the entries that decode to it.  Returns <0 for an over-subscribed set, >0 for an incomplete
// one, 0 for a complete one (puff's `left`); TABLES_FULL when sub[] cannot hold the block's second-level tables (the
// block is then the host's, like any block the decoder does not vouch for); zeros = the number of unused symbols.
// depth: 1 << ROOT bytes of scratch (how many bits each root prefix's second-level table takes).
constexpr int TABLES_FULL = -1000;
template <int ROOT, bool DIST>
__device__ int build_tables(WaveLds& S, const uint8_t* lens, int n, uint32_t* root, uint8_t* depth, int& sub_used, int& zeros, int lane) {
    if (lane < 16) S.hdr.cnt[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += LANES) atomicAdd(&S.hdr.cnt[lens[s]], 1u);
    for (int t = lane; t < (1 << ROOT); t += LANES) { root[t] = BAD_SYMBOL; depth[t] = 0; }
    __syncthreads();
    int c[MAXBITS + 1];
#pragma unroll
    for (int l = 0; l <= MAXBITS; ++l) c[l] = (int)S.hdr.cnt[l];
    zeros = c[0];
    int left = 1;
#pragma unroll
    for (int l = 1; l <= MAXBITS; ++l) {
        left <<= 1;
        left -= c[l];
        if (left < 0) return left;
    }
    int first[MAXBITS + 2];                                // the first code of every length
    first[1] = 0;
#pragma unroll
    for (int l = 1; l <= MAXBITS; ++l) first[l + 1] = (first[l] + c[l]) << 1;
    // the symbol's code, its bits in the order the stream has them (bit 0 first)
    auto code_of = [&](int l, int (&seen)[MAXBITS + 1]) {
        int code = 0;
#pragma unroll
        for (int L = 1; L <= MAXBITS; ++L) {
            if (c[L] == 0) continue;
            const uint64_t m = __builtin_amdgcn_ballot_w64(l == L);
            if (l == L) code = first[L] + seen[L] + lanes_below(m);
            seen[L] += (int)__popcll(m);
        }
        return l > 0 ? (int)(__builtin_bitreverse32((uint32_t)code) >> (32 - l)) : 0;
    };
    // pass 1: codes of at most ROOT bits fill the root table; longer ones say how deep their prefix's table has to be
    // (a length at a time: the lanes that store to one prefix's depth in one instruction store the same value)
    int seen[MAXBITS + 1];
#pragma unroll
    for (int l = 0; l <= MAXBITS; ++l) seen[l] = 0;
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int s = s0 + lane;
        const int l = s < n ? (int)lens[s] : 0;
        const int rev = code_of(l, seen);
        if (l > 0 && l <= ROOT) {
            const uint32_t e = DIST ? entry_D(s, l) : entry_L(s, l);
            for (int i = rev; i < (1 << ROOT); i += 1 << l) root[i] = e;
        }
#pragma unroll
        for (int L = ROOT + 1; L <= MAXBITS; ++L) {
            if (c[L] == 0) continue;
            if (l == L) {
                uint8_t& d = depth[rev & ((1 << ROOT) - 1)];
                d = (uint8_t)max((int)d, L - ROOT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (one wavefront: its LDS operations are carried out in order)
        }
    }
    __syncthreads();
    // the second-level tables one behind the other in sub[]
    constexpr int PER_LANE = ((1 << ROOT) + LANES - 1) / LANES;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {
        const int r = lane * PER_LANE + k;
        if (r < (1 << ROOT) && depth[r] != 0) mine += 1 << depth[r];
    }
    const int incl = wave_incl_scan(mine);
    const int total = __builtin_amdgcn_readlane(incl, LANES - 1);
    if (sub_used + total > SUB_CAP) return TABLES_FULL;
    int at = sub_used + incl - mine;
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {
        const int r = lane * PER_LANE + k;
        if (r < (1 << ROOT) && depth[r] != 0) {
            root[r] = LINK | (uint32_t)depth[r] << 16 | (uint32_t)at;
            at += 1 << depth[r];
        }
    }
    for (int t = lane; t < total; t += LANES) S.sub[sub_used + t] = BAD_SYMBOL;
    __syncthreads();
    // pass 2: the long codes' entries
    if (total > 0) {
#pragma unroll
        for (int l = 0; l <= MAXBITS; ++l) seen[l] = 0;
        for (int s0 = 0; s0 < n; s0 += LANES) {
            const int s = s0 + lane;
            const int l = s < n ? (int)lens[s] : 0;
            const int rev = code_of(l, seen);
            if (l > ROOT) {
                const uint32_t link = root[rev & ((1 << ROOT) - 1)];
                const int bits = (int)((link >> 16) & 15u), base = (int)(link & 0xFFFFu);
                const uint32_t e = DIST ? entry_D(s, l) : entry_L(s, l);
                for (int i = rev >> ROOT; i < (1 << bits); i += 1 << (l - ROOT)) S.sub[base + i] = e;
            }
        }
    }
    sub_used += total;
    __syncthreads();
    return zeros == n ? 0 : left;        // no codes at all: complete, nothing decodes (as puff and zlib have it)
}

// an entry with its LINK followed: `behind` = the stream's bits behind the root's
__device__ __forceinline__ uint32_t follow_link(const WaveLds& S, uint32_t e, uint32_t behind) {
    const uint32_t at = (e & 0xFFFFu) + (behind & ~(~0u << ((e >> 16) & 15u)));
    const uint32_t e2 = S.sub[min(at, (uint32_t)(SUB_CAP - 1))];       // (every lane looks: the ones without a link anywhere in range)
    return (e >> 30) == 3u ? e2 : e;
}

// a match from its two entries: length's extra bits, distance's extra bits, everything packed
__device__ __forceinline__ uint32_t pack_match(uint64_t view, uint32_t eL, uint32_t eD) {
    const uint32_t lx = (uint32_t)(view >> ((eL >> 16) & 15u)) & ~(~0u << ((eL >> 20) & 7u));
    const uint64_t v2 = view >> ((eL >> P_BITS) & 63u);
    const uint32_t dx = (uint32_t)(v2 >> ((eD >> 16) & 15u)) & ~(~0u << ((eD >> 20) & 15u));
    return ((eL & 0xFF8000FFu) + lx) + (eD & 0xFE000000u) + (((eD & 0x7FFFu) + dx) << P_DIST);
}

// the complete symbol that starts at this lane's bit of the stream (57 valid bits in view), packed.  No branches: every
// lane makes all four look-ups (the later ones with whatever bits its earlier entries say follow -- in range by
// construction) and selects.
__device__ __forceinline__ uint32_t symbol_at(uint64_t view, const WaveLds& S) {
    const uint32_t eL = follow_link(S, S.rootL[(uint32_t)view & ((1u << ROOTL) - 1u)], (uint32_t)(view >> ROOTL));
    const uint64_t v2 = view >> ((eL >> P_BITS) & 63u);
    const uint32_t eD = follow_link(S, S.rootD[(uint32_t)v2 & ((1u << ROOTD) - 1u)], (uint32_t)(v2 >> ROOTD));
    const uint32_t m = pack_match(view, eL, eD);
    const bool is_match = ((eL >> P_KIND) & 3u) == K_MATCH;
    return is_match ? ((int32_t)eD < 0 ? BAD_SYMBOL : m) : eL;        // (BAD entries of either table pass through)
}

// Executes the queue: lane k holds symbol k (k < nsym).  Returns 0, or -1 when the output or a distance is out of range.
__device__ __forceinline__ int run_queue(uint8_t* o, int olen, int& opos, uint32_t q, int nsym, int lane) {
    const bool valid = lane < nsym;
    const bool is_match = valid && ((q >> P_KIND) & 3u) == K_MATCH;
    const int val = (int)(q & 255u);
    const int dist = (int)((q >> P_DIST) & 0x7fffu) + 1;
    const int len = valid ? (is_match ? val + 3 : 1) : 0;
    const int incl = wave_incl_scan(len);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (opos + total > olen) return -1;
    const int dst = opos + incl - len;
    if (__builtin_amdgcn_ballot_w64(is_match && dist > dst) != 0) return -1;
    if (valid && !is_match) o[dst] = (uint8_t)val;
    // a match whose source ends before the batch begins depends on nothing in the batch: those of at most 64 bytes copy
    // themselves, 16 bytes per trip, all of them side by side (one memory round trip per trip for the wavefront; the
    // destinations of a batch are consecutive, so the stores fall into few cache lines)
    const bool own = is_match && dst - dist + len <= opos && len <= 64;
    if (own) {
        const uint8_t* src = o + dst - dist;
        uint8_t* d = o + dst;
        for (int left = len; left > 0; left -= 16, src += 16, d += 16) {
            const uint64_t lo = *reinterpret_cast<const U64*>(src);
            if (left >= 8) {
                const uint64_t hi = *reinterpret_cast<const U64*>(src + 8);
                const int n = min(left, 16), sh = (n - 8) * 8;     // bytes [n - 8, n) of hi:lo
                const uint64_t tail = sh == 0 ? lo : (sh == 64 ? hi : (lo >> sh) | (hi << (64 - sh)));
                *reinterpret_cast<U64*>(d) = lo;
                *reinterpret_cast<U64*>(d + n - 8) = tail;         // (overlaps the first store: exactly n bytes are written)
            } else if (left >= 4) {
                *reinterpret_cast<U32*>(d) = (uint32_t)lo;
                *reinterpret_cast<U32*>(d + left - 4) = (uint32_t)(lo >> ((left - 4) * 8));
            } else {
                if (left >= 2) *reinterpret_cast<U16*>(d) = (uint16_t)lo;
                if (left != 2) d[left - 1] = (uint8_t)(lo >> ((left - 1) * 8));
            }
        }
    }
    // the others in stream order, 64 bytes per step by all lanes (a wavefront's memory operations are carried out in
    // order: a step reads what earlier steps, and the stores above, wrote)
    uint64_t rest = __builtin_amdgcn_ballot_w64(is_match && !own);
    while (rest != 0) {
        const int k = (int)__builtin_ctzll(rest);
        rest &= rest - 1;
        const int L = __builtin_amdgcn_readlane(len, k), D = __builtin_amdgcn_readlane(dist, k);
        uint8_t* t = o + __builtin_amdgcn_readlane(dst, k);
        if (D >= LANES) {
            for (int j = lane; j < L; j += LANES) t[j] = t[j - D];
        } else {                                           // the D bytes before the match, repeated
            const int r = lane % D;
            const int step = LANES % D;
            int m = r;
            for (int j = lane; j < L; j += LANES) {
                t[j] = t[m - D];
                m += step;
                if (m >= D) m -= D;
            }
        }
    }
    opos += total;
    return 0;
}

// ---- CRC-32 of a block's inflated bytes (the BGZF trailer's check), by the wavefront that wrote them -------------------
// 64 lanes take 64 equal chunks of the block, padded IN FRONT with zero bytes to 64 * C bytes, C a power of two (a CRC
// register that is still zero stays zero over zero bytes, so the padding changes nothing; the register is set to all
// ones where the data begins, as the standard has it).  Slice-by-4 tables in LDS; the 64 registers are folded by a
// tree, crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B), whose shifts x^(8 C 2^level) come from a table.
constexpr uint32_t CRC_POLY = 0xEDB88320u;
__host__ __device__ constexpr uint32_t multmodp(uint32_t a, uint32_t b) {   // a * b mod P, x^0 at bit 31 (zlib's convention)
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        p ^= ((a >> i) & 1u) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}
struct CrcTables {
    uint32_t t[4][256];     // slice-by-4
    uint32_t x8n[24];       // x^(8 * 2^k) mod P
};
constexpr CrcTables make_crc_tables() {
    CrcTables c = {};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ ((r & 1u) ? CRC_POLY : 0u);
        c.t[0][i] = r;
    }
    for (int k = 1; k < 4; ++k)
        for (uint32_t i = 0; i < 256; ++i) c.t[k][i] = (c.t[k - 1][i] >> 8) ^ c.t[0][c.t[k - 1][i] & 255u];
    uint32_t p = 0x40000000u;                          // x^1
    for (int k = 0; k < 3; ++k) p = multmodp(p, p);    // x^8
    for (int k = 0; k < 24; ++k) { c.x8n[k] = p; p = multmodp(p, p); }
    return c;
}
__constant__ CrcTables CRC = make_crc_tables();

__device__ __forceinline__ uint32_t crc_byte(const uint32_t* T, uint32_t crc, uint32_t byte) { return T[(crc ^ byte) & 255u] ^ (crc >> 8); }
__device__ __forceinline__ uint32_t crc_word(const uint32_t* T, uint32_t crc, uint32_t word) {
    crc ^= word;
    return T[768 + (crc & 255u)] ^ T[512 + ((crc >> 8) & 255u)] ^ T[256 + ((crc >> 16) & 255u)] ^ T[crc >> 24];
}

__device__ uint32_t block_crc(WaveLds& S, const uint8_t* o, int olen, int lane) {
    if (olen == 0) return 0u;
    uint32_t* T = reinterpret_cast<uint32_t*>(S.rootL);      // (the Huffman tables are dead by now: rootL, rootD and sub lie one behind the other)
    static_assert(offsetof(WaveLds, queue) >= 4 * 256 * sizeof(uint32_t), "the CRC tables go where the Huffman tables were");
    __syncthreads();
    for (int k = lane; k < 1024; k += LANES) T[k] = CRC.t[k >> 8][k & 255];
    __syncthreads();
    int lg = 2;                                               // C = 2^lg >= olen / 64
    while ((LANES << lg) < olen) ++lg;
    const int C = 1 << lg;
    int d = lane * C - (LANES * C - olen);                    // the lane's first byte in data coordinates
    const int dend = d + C;
    uint32_t crc = 0;
    if (dend > 0) {
        if (d <= 0) { d = 0; crc = 0xFFFFFFFFu; }             // the data begins inside this lane's chunk
        while (d < dend && ((dend - d) & 15) != 0) { crc = crc_byte(T, crc, o[d]); ++d; }
        for (; d < dend; d += 16) {
            const uint32_t w0 = *reinterpret_cast<const U32*>(o + d), w1 = *reinterpret_cast<const U32*>(o + d + 4),
                           w2 = *reinterpret_cast<const U32*>(o + d + 8), w3 = *reinterpret_cast<const U32*>(o + d + 12);
            crc = crc_word(T, crc, w0);
            crc = crc_word(T, crc, w1);
            crc = crc_word(T, crc, w2);
            crc = crc_word(T, crc, w3);
        }
    }
    for (int l = 0; l < 6; ++l) {
        const uint32_t right = (uint32_t)__shfl_down((int)crc, 1 << l, LANES);
        if ((lane & ((2 << l) - 1)) == 0) crc = multmodp(CRC.x8n[lg + l], crc) ^ right;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)crc) ^ 0xFFFFFFFFu;
}

#ifdef INFLATE_PROF     // cycle counters per phase and block (tools/inflate_prof.hip); the product build has none of this
__device__ unsigned long long* g_prof;
struct Prof {
    unsigned long long t, t0, acc[8];
    __device__ void start() { t = t0 = clock64(); for (int k = 0; k < 8; ++k) acc[k] = 0; }
    __device__ void mark(int k) { const unsigned long long now = clock64(); acc[k] += now - t; t = now; }
    __device__ void count(int k, int n) { acc[k] += n; }
    __device__ void out(int g, int lane) { if (lane == 0 && g_prof) { acc[7] = clock64() - t0; for (int k = 0; k < 8; ++k) g_prof[(size_t)g * 8 + k] = acc[k]; } }
};
#define SYMBOLS_FN __device__ __forceinline__
#else
struct Prof {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void count(int, int) {}
    __device__ __forceinline__ void out(int, int) {}
};
// a function of its own, really called: the header code around it (tables built from uniform arrays, the run-length decoder)
// is large and cold, and inlined into one loop nest with it the compiler shuffled its state through the hot loop
#define SYMBOLS_FN __device__ __noinline__
#endif

struct SymbolsEnd { int rc, opos, bit; };

// Across a real call every argument arrives in vector registers and pointers lose their address space: the callee
// says again that they are wave-uniform (scalar registers, scalar branches) and global (global_load, not flat_load).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
typedef const uint8_t __attribute__((address_space(1)))* GlobalBytesIn;
typedef uint8_t __attribute__((address_space(1)))* GlobalBytes;
template <typename G, typename T>
__device__ __forceinline__ G uniform_global(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint64_t u = (uint64_t)(uint32_t)uniform((int)(uint32_t)v) | (uint64_t)(uint32_t)uniform((int)(uint32_t)(v >> 32)) << 32;
    return (G)(T*)u;
}

// The symbols of one deflate block from bit `bit` of the payload on (tables in S): decodes and executes them until the
// end-of-block code.  rc 0 / -1; opos and bit move on.
SYMBOLS_FN SymbolsEnd decode_symbols(WaveLds& S, const uint8_t* p8_, int nbytes, uint8_t* o_, int olen, int opos, int bit, Prof& P) {
    // (cast back to plain pointers: the compiler follows the address space through them to every load and store)
    const uint8_t* p8 = (const uint8_t*)uniform_global<GlobalBytesIn>(p8_);
    uint8_t* o = (uint8_t*)uniform_global<GlobalBytes>(o_);
    nbytes = uniform(nbytes); olen = uniform(olen); opos = uniform(opos); bit = uniform(bit);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // window = the 64 bit offsets bit0 .. bit0 + 63; pos = where the next symbol starts, relative to bit0
    int bit0 = bit, pos = 0, nsym = 0, end = 0, rc = 0;
    uint64_t raw = *reinterpret_cast<const U64*>(p8 + ((bit0 + lane) >> 3));
    uint64_t raw1 = *reinterpret_cast<const U64*>(p8 + ((bit0 + LANES + lane) >> 3));
    while (!end) {
        if (bit0 > nbytes * 8) { rc = -1; break; }         // a symbol would start behind the payload
        const uint64_t view = raw >> ((bit0 + lane) & 7);
        raw = raw1;
        raw1 = *reinterpret_cast<const U64*>(p8 + ((bit0 + 2 * LANES + lane) >> 3));   // the window after the next, early
        const uint32_t sp = symbol_at(view, S);
        P.mark(1);
        // the walk: from symbol start to symbol start, on the scalar unit
        uint64_t starts = 0;
        uint32_t e;
        // (four steps per trip of the loop: a taken branch costs a lone wavefront as much as the step itself)
#define WALK_STEP                                                    \
            e = (uint32_t)__builtin_amdgcn_readlane((int)sp, pos);   \
            asm("s_bitset1_b64 %0, %1" : "+s"(starts) : "s"(pos));   \
            pos += (int)(e >> P_BITS);
        for (;;) {
            WALK_STEP
            if (pos >= LANES) break;
            WALK_STEP
            if (pos >= LANES) break;
            WALK_STEP
            if (pos >= LANES) break;
            WALK_STEP
            if (pos >= LANES) break;
        }
#undef WALK_STEP
        P.mark(2);
        P.count(6, 1);
        const uint32_t kind = (e >> P_KIND) & 3u;
        if (kind >= K_END) {                               // the last symbol of the block (or nothing decodable): not queued
            if (kind == K_BAD) { rc = -1; break; }
            const int at = pos - LANES;
            starts &= ~(1ull << at);
            bit = bit0 + at + (int)(e & 255u);
            end = 1;
        }
        // the lanes that hold a real symbol append it to the queue, in stream order
        if ((starts >> lane) & 1ull) S.queue[nsym + lanes_below(starts)] = sp;
        nsym += (int)__popcll(starts);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (one wavefront: its LDS operations are carried out in order)
        P.mark(3);
        while (nsym >= LANES || (end && nsym > 0)) {
            const int n = min(nsym, LANES);
            const uint32_t q = S.queue[lane];
            const uint32_t q2 = S.queue[LANES + lane];
            if (run_queue(o, olen, opos, q, n, lane) != 0) { rc = -1; end = 1; break; }
            nsym -= n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < nsym) S.queue[lane] = q2;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            P.mark(4);
        }
        pos -= LANES;
        bit0 += LANES;
    }
    return SymbolsEnd{rc, opos, bit};
}

#ifndef INFLATE_WAVES
#define INFLATE_WAVES 7
#endif
__global__ __launch_bounds__(LANES, INFLATE_WAVES) void inflate_kernel(const uint32_t* __restrict__ comp, const int64_t* __restrict__ comp_off,
                                                        uint8_t* out, const int64_t* __restrict__ out_off, int first_block,
                                                        int32_t* __restrict__ status, uint32_t* __restrict__ crc_out) {
    __shared__ WaveLds S;
    const int lane = threadIdx.x;
    const int g = first_block + blockIdx.x;
    const int64_t c0 = comp_off[g], c1 = comp_off[g + 1];
    uint8_t* o = out + out_off[g];
    const int olen = (int)(out_off[g + 1] - out_off[g]);
    const uint32_t* p = comp + (c0 >> 2);
    const uint8_t* p8 = reinterpret_cast<const uint8_t*>(p);
    const int nbytes = (int)(c1 - c0);
    const int nwords = (nbytes + 3) >> 2;
    int bit = 0, opos = 0, rc = nbytes > 0 ? 0 : -1, last = 0;
    Prof P;
    P.start();
    while (rc == 0 && !last) {
        // ---- a deflate block header ----
        UBits b;
        b.start(p, nwords, bit);
        last = (int)b.get(1);
        const int type = (int)b.get(2);
        if (type == 0) {                                   // stored: LEN, ~LEN on the next byte boundary, then the bytes
            b.skip(b.cnt & 7);
            const uint32_t len = b.get(16), nlen = b.get(16);
            const int from = b.pos() >> 3;
            if ((len ^ 0xffffu) != nlen || opos + (int)len > olen || from + (int)len > nwords * 4) { rc = -1; break; }
            for (int j = lane; j < (int)len; j += LANES) o[opos + j] = p8[from + j];
            opos += (int)len;
            bit = (from + (int)len) * 8;
            continue;
        }
        if (type == 3) { rc = -1; break; }
        int nlen = MAXL, ndist = MAXD;                     // (the fixed codes are complete over 288 / 32 symbols: the
        if (type == 1) {                                   //  symbols no stream may use are refused where they turn up)
            for (int s = lane; s < MAXL; s += LANES) S.hdr.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            if (lane < MAXD) S.hdr.lens[MAXL + lane] = 5;
        } else {
            nlen = (int)b.get(5) + 257;
            ndist = (int)b.get(5) + 1;
            const int ncode = (int)b.get(4) + 4;
            if (nlen > 286 || ndist > 30) { rc = -1; break; }
            // the code-length code: lane s holds the length of its symbol s (19 symbols)
            int mycl = 0;
            for (int k = 0; k < ncode; ++k) {
                const int v = (int)b.get(3);
                if (lane == CL_ORDER[k]) mycl = v;
            }
            int left = 1, offs = 0, myrank = 0;
            int cc[8];
#pragma unroll
            for (int L = 1; L <= 7; ++L) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(mycl == L);
                cc[L] = (int)__popcll(m);
                if (mycl == L) myrank = offs + lanes_below(m);
                offs += cc[L];
                left = (left << 1) - cc[L];
                if (left < 0) break;
            }
            if (left != 0) { rc = -1; break; }             // a complete code is required (as zlib does)
            if (mycl != 0) S.hdr.clsym[myrank] = (uint8_t)lane;
            __syncthreads();
            for (int t = lane; t < 128; t += LANES) {
                int code = 0, first = 0, index = 0, found = -1, flen = 0;
#pragma unroll
                for (int len = 1; len <= 7; ++len) {
                    code |= (t >> (len - 1)) & 1;
                    const int count = cc[len];
                    if (found < 0 && code - count < first) { found = index + (code - first); flen = len; }
                    index += count;
                    first += count;
                    first <<= 1;
                    code <<= 1;
                }
                S.hdr.clfast[t] = found < 0 ? (uint8_t)0 : (uint8_t)(S.hdr.clsym[found] | flen << 5);
            }
            __syncthreads();
            // the nlen + ndist code lengths, run-length coded
            int idx = 0, prev = 0;
            const int total = nlen + ndist;
            while (idx < total) {
                const uint32_t e = S.hdr.clfast[b.peek(7)];
                if (e == 0) { rc = -1; break; }
                b.skip((int)(e >> 5));
                const int sym = (int)(e & 31u);
                if (sym < 16) {
                    if (lane == 0) S.hdr.lens[idx] = (uint8_t)sym;
                    prev = sym;
                    ++idx;
                } else {
                    int rep;
                    if (sym == 16) {
                        if (idx == 0) { rc = -1; break; }
                        rep = 3 + (int)b.get(2);
                    } else if (sym == 17) { prev = 0; rep = 3 + (int)b.get(3); }
                    else { prev = 0; rep = 11 + (int)b.get(7); }
                    if (idx + rep > total) { rc = -1; break; }
                    for (int j = lane; j < rep; j += LANES) S.hdr.lens[idx + j] = (uint8_t)prev;
                    idx += rep;
                }
            }
            if (rc != 0) break;
            __syncthreads();
            if (S.hdr.lens[256] == 0) { rc = -1; break; }      // no end-of-block code
            // the distance lengths follow the literal/length lengths directly: move them to their own place
            const int dl = lane < ndist ? (int)S.hdr.lens[nlen + lane] : 0;
            __syncthreads();
            if (lane < MAXD) S.hdr.lens[MAXL + lane] = (uint8_t)dl;
        }
        bit = b.pos();
        __syncthreads();
        // (scratch for the tables' construction: the distance table's room while the literal / length code is built, the
        //  literal / length code's lengths -- done with by then -- while the distance code is)
        int zeros, sub_used = 0;
        int err = build_tables<ROOTL, false>(S, S.hdr.lens, nlen, S.rootL, reinterpret_cast<uint8_t*>(S.rootD), sub_used, zeros, lane);
        static_assert(sizeof(S.rootD) >= (1 << ROOTL) && MAXL >= (1 << ROOTD), "scratch for build_tables");
        if (err == TABLES_FULL) { rc = -3; break; }
        if (err < 0 || (err > 0 && nlen - zeros != 1)) { rc = -1; break; }
        err = build_tables<ROOTD, true>(S, S.hdr.lens + MAXL, ndist, S.rootD, S.hdr.lens, sub_used, zeros, lane);
        if (err == TABLES_FULL) { rc = -3; break; }
        if (err < 0 || (err > 0 && ndist - zeros != 1)) { rc = -1; break; }

        P.mark(0);
        // ---- the block's symbols ----
        const SymbolsEnd r = decode_symbols(S, p8, nbytes, o, olen, opos, bit, P);
        rc = r.rc; opos = r.opos; bit = r.bit;
        if (rc != 0) break;
        __syncthreads();                                   // the tables are rebuilt by the next header
    }
    if (rc == 0) {
        if (opos != olen) rc = -2;                         // fewer bytes than the trailer's ISIZE
        else if (bit > nbytes * 8) rc = -1;                // ran past the payload
    }
    P.mark(3);
    if (crc_out) {
        const uint32_t crc = rc == 0 ? block_crc(S, o, olen, lane) : 0u;
        if (lane == 0) crc_out[g] = crc;
    }
    P.mark(5);
    P.out(g, lane);
    if (lane == 0) status[g] = rc;
}
}  // namespace

hipError_t tredgpu_front::launch_inflate(const uint32_t* comp, const int64_t* comp_off, uint8_t* out, const int64_t* out_off, int first_block,
                                         int n_blocks, int32_t* status, uint32_t* crc_out, hipStream_t st) {
    if (n_blocks <= 0) return hipSuccess;
    inflate_kernel<<<n_blocks, LANES, 0, st>>>(comp, comp_off, out, out_off, first_block, status, crc_out);
    return hipGetLastError();
}
