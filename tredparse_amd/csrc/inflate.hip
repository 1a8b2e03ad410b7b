// inflate.hip -- raw DEFLATE (RFC 1951) decoding of many BGZF blocks at once on gfx950.
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

#include "../../include/tredgpu.h"

namespace {

constexpr int LANES = 64;
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

__device__ __forceinline__ int wave_incl_scan(int v) {   // all 64 lanes active
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr 1, 2, 4, 8: prefix inside a 16-lane row
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast 15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast 31 into rows 2 and 3
    return v;
}
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

// ---- the pair walk: PEextractor over the blocks this decoder just wrote ---------------------------------------------------
// tredparse/bam_parser.py:316-369 (PEextractor) as the host's file layer restates it (bamread.cpp: walk_region, PairTable):
// the records of a +-10 kb region in file order; paired, mapped, non-duplicate reads grouped by query name in order of first
// appearance; of every name seen twice the first two records must map +/-; tlen from the soft-clipped ends; a pair that
// spans the tract goes to the target list, any other to the global one.
//
// Four launches per call (walk_chain_par_kernel -> walk_chain_kernel for what that one hands back -> walk_parse_kernel ->
// pair_walk_kernel; the comments at each say what it does and why).  Nothing here reads HBM a record at a time: a lane
// walking the records of a region in global memory paid a miss of 1-2 us for each, and several per record (24 ms for a
// region of 4 000 records) -- the serial chain goes through a 6 KB window in LDS, the parallel one gives every lane ~60
// records, the name table lives in LDS.  What a hash match in that table does NOT prove -- that two names are equal byte
// for byte -- is checked for all pairs at the end by all lanes, and a single mismatch there gives the region back to the
// host, as does anything else out of the ordinary: a block the plan does not hold or the decoder rejected or whose CRC-32
// is not its trailer's, a record that makes no sense, more names than the table holds.  The host then walks that region
// itself, as it does without these kernels, and reports what is wrong with the file.
struct WalkView {
    const uint8_t* out; const int64_t* ooff;          // the decoder's output and its block offsets
    const int32_t* bstatus; const uint32_t* bcrc;     // what the decoder said about each block
    const uint32_t* xcrc; const int64_t* bcoff; const int32_t* bclen;   // from the file: trailer CRC, compressed offset / length
    int64_t out_end;                                  // bytes of `out` that may be read
};
struct WalkPair { int64_t name_at, name2_at; int32_t a_pos, a_lead, b_end, b_trail; uint16_t name_len; uint8_t a_rev, b_rev, complete, pad[3]; };
static_assert(sizeof(WalkPair) == 40, "WalkPair layout");
constexpr int WALK_PAIR_CAP = 8192;               // names per region at most (a +-10 kb window at 30x holds ~2 100) ...
constexpr int WALK_PAIR_CAP_SMALL = 4096;         // ... and what a launch whose regions are all short is given: 38 instead of
                                                  // 70 KB of LDS per wavefront, so that other kernels' workgroups -- the
                                                  // decoder's, the genotyping kernels' of the other driver processes -- still
                                                  // find LDS on the CUs a walk occupies (two walks of 70 KB nearly fill a CU's 160 KB)
constexpr int WALK_WINDOW = 6144;                 // bytes of the block stream in LDS
enum { WALK_OK = 0, WALK_NOT_PLANNED = 1, WALK_BAD_BLOCK = 2, WALK_BAD_RECORD = 3, WALK_TABLE_FULL = 4, WALK_NO_END = 5, WALK_POOL_FULL = 6,
       WALK_TAG_CLASH = 7 };

// The workgroup is one wavefront, and a wavefront's LDS instructions are carried out in the order they were issued: what
// lane 0 writes is there when the next instruction of any lane reads it.  No s_barrier is needed -- and __syncthreads()
// must not be used in the record loop: it also waits for every global store before it (vmcnt(0)), 1-2 us after each of the
// pair entries lane 0 writes (measured: 2.5 us per record with it).  This only keeps the compiler from moving LDS
// accesses across the point.
__device__ inline void walk_lds_order() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// (LDS pointers keep their address space through the struct: as plain pointers they became flat_load / flat_store)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef uint32_t walk_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) walk_u32x4 lds_u128;
struct WalkLds {                                  // (views into the launch's dynamic LDS: walk_lds_bytes(cap))
    lds_u32* table;                               // 2 * cap slots, open addressing at a load below one half.  0: free; else
                                                  // tag << 15 | records under the tag so far (saturates at 3) << 13 | pair index, tag != 0
    lds_u8* window;                               // WALK_WINDOW bytes, 16-byte aligned
    int cap; uint32_t mask;
};
constexpr size_t walk_lds_bytes(int cap) { return (size_t)cap * 16; }    // (pair_walk_kernel: the table alone, 2 * cap slots of 64 bits)

// Every lane holds the same value: say so (v_readfirstlane), and what is computed from it is computed once, on the
// scalar unit, with scalar branches -- not 64 times on the vector unit with the exec mask rebuilt at every `if`.
__device__ inline uint32_t walk_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ inline int64_t walk_uniform64(int64_t x) { return (int64_t)((uint64_t)walk_uniform((uint32_t)((uint64_t)x >> 32)) << 32 | walk_uniform((uint32_t)x)); }

__device__ inline uint64_t walk_lane64(uint64_t x, int j) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), j) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, j);
}

// the block stream through the LDS window
struct WalkReader {
    const uint8_t* out; int64_t out_end; WalkLds* S; int64_t base; int lane;
    __device__ void fill(int64_t at) {
        base = at & ~(int64_t)15;
        walk_lds_order();                                        // (earlier reads of the window are done)
        for (int q = 0; q < WALK_WINDOW / (LANES * 16); ++q) {
            const int o = (q * LANES + lane) * 16;
            if (base + o + 16 <= out_end) *(lds_u128*)(S->window + o) = *(const walk_u32x4*)(out + base + o);
        }
        walk_lds_order();
    }
    // the next window, fetched while the batch at hand is parsed and resolved: the loads are issued (prefetch) once the
    // chain knows where the next batch starts, and land in LDS (commit) when nothing reads the old window any more
    walk_u32x4 ahead[WALK_WINDOW / (LANES * 16)];
    int64_t ahead_base;
    __device__ void prefetch(int64_t at) {
        ahead_base = at & ~(int64_t)15;
        for (int q = 0; q < WALK_WINDOW / (LANES * 16); ++q) {
            const int o = (q * LANES + lane) * 16;
            if (ahead_base + o + 16 <= out_end) ahead[q] = *(const walk_u32x4*)(out + ahead_base + o);
        }
    }
    __device__ void commit() {
        walk_lds_order();
        for (int q = 0; q < WALK_WINDOW / (LANES * 16); ++q) {
            const int o = (q * LANES + lane) * 16;
            if (ahead_base + o + 16 <= out_end) *(lds_u128*)(S->window + o) = ahead[q];
        }
        base = ahead_base;
        walk_lds_order();
    }
    __device__ bool inside(int64_t at, int n) const { return at >= base && at + n <= base + WALK_WINDOW; }
    // (per lane: its own address)
    __device__ uint32_t vu8(int64_t at) const { return inside(at, 1) ? S->window[at - base] : out[at]; }
    __device__ uint32_t vu16(int64_t at) const { return vu8(at) | (vu8(at + 1) << 8); }
    __device__ uint32_t vu32(int64_t at) const { return vu16(at) | (vu16(at + 2) << 16); }
    // (every lane the same address)
    __device__ uint32_t u8(int64_t at) const { return walk_uniform(inside(at, 1) ? S->window[at - base] : out[at]); }
    __device__ uint32_t u16(int64_t at) const { return u8(at) | (u8(at + 1) << 8); }
    __device__ uint32_t u32(int64_t at) const {
        if (inside(at, 4)) { const lds_u8* p = S->window + (at - base); return walk_uniform((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
        uint32_t v; __builtin_memcpy(&v, out + at, 4); return walk_uniform(v);
    }
};

__device__ inline bool walk_block_ok(const WalkView& v, int k) { return v.bstatus[k] == 0 && v.bcrc[k] == v.xcrc[k]; }

// Where the walk stands in the file: block k (its size, compressed offset and length held in registers: the tables are
// read once per block, not once per record -- a load in the record loop would also wait for the stores before it) and
// the offset in it.
struct WalkCursor {
    int k; int64_t upos, size, first, coff, clen; uint64_t here, next;     // (here / next: this block's and the next one's virtual offsets)
    __device__ int enter(const WalkView& v, int block) {
        k = block;
        if (!walk_block_ok(v, k)) return WALK_BAD_BLOCK;
        first = walk_uniform64(v.ooff[k]);
        size = walk_uniform64(v.ooff[k + 1]) - first;
        coff = walk_uniform64(v.bcoff[k]);
        clen = (int64_t)walk_uniform((uint32_t)v.bclen[k]);
        here = (uint64_t)coff << 16;
        next = (uint64_t)(coff + clen) << 16;
        return WALK_OK;
    }
    __device__ uint64_t tell() const { return upos >= size ? next : here | (uint64_t)upos; }   // bamread.cpp bg_tell
    // the next n bytes of the file: blocks that follow each other in the file follow each other in `out`, so the bytes lie
    // in one piece at *addr; the position moves as bamread.cpp's bg_read moves it
    __device__ int take(const WalkView& v, const tredgpu_walk_task& T, int64_t n, int64_t* addr) {
        *addr = -1;
        while (n > 0) {
            if (upos >= size) {
                if (k + 1 >= T.block_end || coff + clen != v.bcoff[k + 1]) return WALK_NOT_PLANNED;   // (or the end of the file)
                const int rc = enter(v, k + 1);
                if (rc) return rc;
                upos = 0;
            }
            if (*addr < 0) *addr = first + upos;
            const int64_t piece = n < size - upos ? n : size - upos;
            upos += piece;
            n -= piece;
        }
        return WALK_OK;
    }
};

// The region's walk, taken apart (round 4: one wavefront walked, parsed and paired a region's ~4 000 records batch by
// batch: 4.06 ms for the 480 regions of 16 samples, on half of the chip's SIMDs, every step waiting for the one before):
//   chain    where every record lies (WalkRec).  walk_chain_par_kernel: 64 lanes per region, each from a guessed record
//            start that the lane before it proves; walk_chain_kernel, one wavefront following the length words through
//            the LDS window, for the regions the lanes hand back;
//   parse    walk_parse_kernel, one LANE per record, every record of every region of the call side by side (~2 million of
//            them for 16 samples: the chip is full): fields, CIGAR (end on the reference, soft clips), name hash ->
//            WalkFields, 32 bytes per record;
//   resolve  pair_walk_kernel, eight wavefronts per region: PairTable::add by LDS atomics (first and second record per
//            name, pairs ranked by ballot and prefix sum), then PairTable::finish.
struct WalkRec { int64_t a0; uint64_t at, after; };             // where the record's length word lies in `out`; the virtual offsets of the record and of what follows it
struct WalkFields { uint32_t h; int32_t rtid, rpos, rend, lead, trail; uint16_t flag, nlen; uint32_t bad; };
static_assert(sizeof(WalkRec) == 24 && sizeof(WalkFields) == 32, "record tuples");
struct WalkChained { int32_t status, n, mode, klo, khi, pad; };   // per region: how the chain ended, records listed; mode 1: listed by
                                                                  // walk_chain_par_kernel (the region's blocks lie in [klo, khi)), 0: by
                                                                  // walk_chain_kernel (pad: why the lanes handed the region back)

constexpr int CHAIN_BATCH = 64;

// The next (up to) CHAIN_BATCH records of chunk `ch` from the cursor on: lane j ends up with record j's place (where its
// length word lies in `out`, its virtual offset, the virtual offset of what follows it).  chunk_done: the chunk's end or the
// first record beyond the region was reached (that record is not listed); status: why the walk cannot go on.
__device__ __forceinline__ int chain_batch(const WalkView& v, const tredgpu_walk_task& T, const tredgpu_walk_chunk& ch, WalkCursor& cur,
                                           WalkReader& rd, WalkLds& S, int lane, int64_t& my_a0, uint64_t& my_at, uint64_t& my_after,
                                           bool& chunk_done, int& status) {
    int nb = 0, rc;
    while (nb < CHAIN_BATCH) {
        // ---- the fast path: records whose length word, contig and position lie in this block AND in the window, and
        //      that end inside the block -- 32-bit arithmetic on (offset in the block, offset in the window) only, one
        //      LDS access per record (~40 instructions; a lone wavefront issues one every ~5 cycles, and the general
        //      step below, 64-bit throughout, took ~250 of them: 2.7 ms per region of 4 000 records) ----
        {
            const int bsz = (int)cur.size;
            int up = (int)cur.upos;
            int wo = (int)walk_uniform((uint32_t)((cur.first + cur.upos) - rd.base));     // (garbage when far outside: checked below)
            const bool near = cur.first + cur.upos >= rd.base && cur.first + cur.upos < rd.base + WALK_WINDOW;
            // where in this block tell() reaches the chunk's end (tell() = here | upos inside a block)
            const uint64_t endv = ch.end_voffset;
            const int up_end = (endv >> 16) == (uint64_t)cur.coff ? (int)(endv & 0xFFFFu) : (endv > cur.here ? 0x7FFFFFFF : 0);
            bool stop = false;
            if (near) {
                while (nb < CHAIN_BATCH && up + 12 <= bsz && up < up_end && wo + 16 <= WALK_WINDOW) {
                    const lds_u32* p = (const lds_u32*)(S.window + (wo & ~3));
                    const uint32_t x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
                    const uint32_t by = (uint32_t)(wo & 3);
                    const int32_t size = (int32_t)walk_uniform(__builtin_amdgcn_alignbyte(x1, x0, by));
                    const int32_t rtid = (int32_t)walk_uniform(__builtin_amdgcn_alignbyte(x2, x1, by));
                    const int32_t rpos = (int32_t)walk_uniform(__builtin_amdgcn_alignbyte(x3, x2, by));
                    const int nxt = up + 4 + size;
                    if (size < 32 || nxt > bsz) break;                    // (the general step decides: a bad record, or one that crosses into the next block)
                    if (rtid > T.tid || (rtid == T.tid && rpos >= T.end)) { stop = true; break; }
                    if (lane == nb) {
                        my_a0 = cur.first + up;
                        my_at = cur.here | (uint64_t)(uint32_t)up;
                        my_after = nxt >= bsz ? cur.next : cur.here | (uint64_t)(uint32_t)nxt;
                    }
                    up = nxt;
                    wo += 4 + size;
                    ++nb;
                }
                cur.upos = up;
            }
            if (stop) { chunk_done = true; break; }
            if (nb >= CHAIN_BATCH) break;
        }
        // ---- the general step: one record, wherever it lies ----
        const uint64_t at = cur.tell();
        if (at >= ch.end_voffset) { chunk_done = true; break; }
        int64_t a0, r;
        if (cur.upos + 4 <= cur.size) { a0 = cur.first + cur.upos; cur.upos += 4; }       // (nearly always)
        else if ((rc = cur.take(v, T, 4, &a0)) != 0) { status = rc; break; }
        if (!rd.inside(a0, 16)) {
            if (rd.ahead_base <= a0 && a0 + 16 <= rd.ahead_base + WALK_WINDOW) rd.commit(); else rd.fill(a0);
            rd.prefetch(rd.base + WALK_WINDOW - 16);       // (the window after this one, while this one is walked)
        }
        const int32_t size = (int32_t)rd.u32(a0);
        if (size < 32) { status = WALK_BAD_RECORD; break; }
        if (cur.upos + size <= cur.size) cur.upos += size;
        else if ((rc = cur.take(v, T, size, &r)) != 0) { status = rc; break; }
        const int32_t rtid = (int32_t)rd.u32(a0 + 4), rpos = (int32_t)rd.u32(a0 + 8);
        if (rtid > T.tid || (rtid == T.tid && rpos >= T.end)) { chunk_done = true; break; }   // beyond the region: the walk over this chunk ends
        if (lane == nb) { my_a0 = a0; my_at = at; my_after = cur.tell(); }
        ++nb;
    }
    return nb;
}

// ---- the chain, 64 lanes at once ------------------------------------------------------------------------------------
// Where a record starts is written in the one before it -- but WHETHER a place is a record's start can be guessed from the
// place itself (a length word that covers the fixed fields, the region's contig, a name that ends in NUL where the head
// says it ends), and a guess can be checked: the chunk's bytes are cut into 64 segments, lane 0 starts at the chunk's
// first record, every other lane at the first place of its segment that looks like a record, and each follows the length
// words up to where the next lane started.  A lane that arrives EXACTLY there has proved the next lane's start (lane 0's is
// true; by induction so are all of them); one that steps over it, meets a length below 32 or leaves the planned blocks has
// not, and then the region is walked by walk_chain_kernel, one record after the other, as before -- so the result is that
// kernel's whatever the bytes are.  A region's chain is ~60 dependent loads per lane instead of ~4 000 steps of one
// wavefront (1.08 ms per launch, a third of the pair walk).
constexpr int PAR_SEG_MIN = 2048;                                  // bytes per lane at least
__device__ inline uint32_t g_u32(const uint8_t* out, int64_t at);
__device__ inline uint32_t g_u16(const uint8_t* out, int64_t at);
__device__ inline uint32_t g_u8(const uint8_t* out, int64_t at);

// the virtual offset bamread.cpp's bg_tell gives at byte `addr` of `out`, blocks [lo, hi) following each other in the file:
// inside a block its offset | the place in it; at a block's end the offset of the block that follows in the file
__device__ inline uint64_t walk_voffset(const WalkView& v, int lo, int hi, int64_t addr) {
    int a = lo, b = hi;                                            // the first k of [lo, hi] with ooff[k] >= addr
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (v.ooff[mid] >= addr) b = mid; else a = mid + 1;
    }
    if (v.ooff[a] == addr) return a < hi ? (uint64_t)v.bcoff[a] << 16 : (uint64_t)(v.bcoff[hi - 1] + v.bclen[hi - 1]) << 16;
    return (uint64_t)v.bcoff[a - 1] << 16 | (uint64_t)(addr - v.ooff[a - 1]);
}

// One chunk of a region by all lanes.  false: not this way (see above).  Else lane j has `cnt` records from address `s` on
// (0 for the lanes behind the one that met the region's end), `first` = how many records the lanes before it have, `total`
// all of them; [klo, khi) grows to hold the chunk's blocks.
struct ParChunk { int64_t s; int cnt, first, total; };
__device__ bool chain_par_chunk(const WalkView& v, const tredgpu_walk_task& T, const tredgpu_walk_chunk& ch, int lane, int min_bytes,
                                ParChunk& out_c, int& klo, int& khi, int* why = nullptr) {
#define PAR_NO(code) do { if (why) *why = (code); return false; } while (0)
    const uint8_t* out = v.out;
    out_c = ParChunk{0, 0, 0, 0};
    const int k0 = ch.begin_block;
    if (k0 < T.block_first || k0 >= T.block_end) PAR_NO(1);
    // the chunk's blocks: k0 .. kend, kend the last planned block that begins at or before the chunk's end; all of them
    // vouched for by the decoder and one behind the other in the file
    const int64_t coff_e = (int64_t)(ch.end_voffset >> 16);
    const int upos_e = (int)(ch.end_voffset & 0xFFFFu);
    int kend = k0 - 1;
    bool fine = true;
    for (int kb = k0; kb < T.block_end; kb += LANES) {
        const int k = kb + lane;
        const bool in = k < T.block_end && v.bcoff[k] <= coff_e;
        kend = max(kend, kb - 1 + (int)__popcll(__ballot(in)));          // (bcoff ascends)
        if (in) {
            fine = fine && walk_block_ok(v, k);
            if (k > k0) fine = fine && v.bcoff[k - 1] + v.bclen[k - 1] == v.bcoff[k];
        }
        if (__ballot(k < T.block_end && !in) != 0) break;
    }
    if (kend < k0) PAR_NO(2);
    if (__ballot(!fine) != 0) PAR_NO(3);
    const int64_t a_lim = v.ooff[kend + 1];                              // what lies behind is not this chunk's
    const int64_t A0 = v.ooff[k0] + ch.begin_upos;
    int64_t A1 = a_lim;
    if (v.bcoff[kend] == coff_e) A1 = min(a_lim, v.ooff[kend] + (int64_t)upos_e);
    // (the chunk goes on where the plan ends: fine when the region's last record comes first -- the plan holds the blocks up
    //  to there --, not when the chain runs off the end)
    const bool open_end = v.bcoff[kend] != coff_e && (uint64_t)(v.bcoff[kend] + v.bclen[kend]) << 16 < ch.end_voffset;
    klo = min(klo, k0); khi = max(khi, kend + 1);
    if (A0 >= A1) { if (open_end) PAR_NO(4); return true; }
    const int64_t len = A1 - A0;
    if (len < min_bytes) PAR_NO(5);                                      // (a few records: the window in LDS is the faster way)
    const int nseg = (int)min((int64_t)LANES, max((int64_t)1, len / PAR_SEG_MIN));
    const int64_t L = (len + nseg - 1) / nseg;
    // ---- where this lane starts ----
    const int64_t INF = (int64_t)1 << 60;
    int64_t s = INF;
    if (lane == 0) s = A0;
    else if (lane < nseg) {
        const int64_t p_end = min(A0 + (lane + 1) * L, A1);
        for (int64_t p = A0 + lane * L; p < p_end && p + 36 <= a_lim; ++p) {
            const int32_t size = (int32_t)g_u32(out, p);
            if (size < 36 || size > (1 << 24) || (int32_t)g_u32(out, p + 4) != T.tid) continue;
            const int32_t rpos = (int32_t)g_u32(out, p + 8), l_seq = (int32_t)g_u32(out, p + 20);
            const int64_t l_name = g_u8(out, p + 12), n_cigar = g_u16(out, p + 16);
            if (rpos < 0 || l_seq < 0 || l_name < 1 || 32 + l_name + 4 * n_cigar + ((int64_t)l_seq + 1) / 2 + l_seq > size) continue;
            if (p + 4 + size > a_lim || g_u8(out, p + 36 + l_name - 1) != 0) continue;
            s = p;
            break;
        }
    }
    // the next lane that has a start (or the chunk's end)
    const uint64_t have = __ballot(s != INF);
    const uint64_t above = lane < 63 ? have >> (lane + 1) : 0;
    const int nextl = above ? lane + 1 + __builtin_ctzll(above) : lane;
    int64_t target = (int64_t)__shfl((unsigned long long)s, nextl, LANES);
    const bool is_last = above == 0;
    if (is_last) target = A1;
    // ---- count: follow the length words from s to target ----
    enum { CLEAN = 0, STOPPED = 1, ANOMALY = 2 };
    int outcome = CLEAN, cnt = 0;
    if (s != INF) {
        int64_t p = s;
        while (p < target) {
            if (p + 12 > a_lim) { outcome = ANOMALY; break; }
            const int32_t size = (int32_t)g_u32(out, p);
            const int32_t rtid = (int32_t)g_u32(out, p + 4), rpos = (int32_t)g_u32(out, p + 8);
            if (size < 32 || p + 4 + (int64_t)size > a_lim) { outcome = ANOMALY; break; }   // (the serial chain says what it is)
            if (rtid > T.tid || (rtid == T.tid && rpos >= T.end)) { outcome = STOPPED; break; }
            ++cnt;
            p += 4 + (int64_t)size;
        }
        if (outcome == CLEAN && p != target && !is_last) outcome = ANOMALY;     // stepped over the next lane's start: a wrong guess
    }
    // the first lane that did not arrive: up to it the chain is the file's
    const uint64_t not_clean = __ballot(outcome != CLEAN);
    const int J = not_clean ? __builtin_ctzll(not_clean) : LANES - 1;
    if (not_clean && __shfl(outcome, J, LANES) == ANOMALY) PAR_NO(6);
    if (!not_clean && open_end) PAR_NO(4);
    const int mycnt = lane <= J ? cnt : 0;
    const int incl = wave_incl_scan(mycnt);
    out_c = ParChunk{s, mycnt, incl - mycnt, __builtin_amdgcn_readlane(incl, 63)};
    return true;
#undef PAR_NO
}

// the same for places that only go up: k = the first block of [lo, hi] that begins at or behind the place, its edges kept
struct VoffCursor {
    int k, lo, hi;
    int64_t edge, before;                              // ooff[k], ooff[k - 1]
    uint64_t vk, vbefore;                              // the virtual offsets of block k's and block k - 1's first byte
    __device__ void load(const WalkView& v) {
        edge = v.ooff[k];
        vk = k < hi ? (uint64_t)v.bcoff[k] << 16 : (uint64_t)(v.bcoff[hi - 1] + v.bclen[hi - 1]) << 16;
        before = k > lo ? v.ooff[k - 1] : 0;
        vbefore = k > lo ? (uint64_t)v.bcoff[k - 1] << 16 : 0;
    }
    __device__ void start(const WalkView& v, int lo_, int hi_, int64_t addr) {
        lo = lo_; hi = hi_;
        int a = lo, b = hi;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (v.ooff[mid] >= addr) b = mid; else a = mid + 1;
        }
        k = a;
        load(v);
    }
    __device__ uint64_t at(const WalkView& v, int64_t addr) {
        while (k < hi && edge < addr) { ++k; load(v); }
        return edge == addr ? vk : vbefore | (uint64_t)(addr - before);
    }
};

__global__ void __launch_bounds__(LANES) walk_chain_par_kernel(WalkView v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks,
                                                               const int64_t* rec_base, WalkRec* recs, WalkChained* chained) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const tredgpu_walk_task T = tasks[t];
    const uint8_t* out = v.out;
    WalkRec* mine = recs + rec_base[t];
    const int64_t cap = rec_base[t + 1] - rec_base[t];
    if (T.n_chunks < 0) {
        if (lane == 0) chained[t] = WalkChained{WALK_NOT_PLANNED, 0, 1, 0, 0, 0};
        return;
    }
    int64_t n = 0;
    int klo = T.block_end, khi = T.block_first, why = 7;      // (why a region is left to the serial chain: WalkChained.pad)
    bool give_up = false;
    for (int c = 0; c < T.n_chunks && !give_up; ++c) {
        ParChunk pc;
        int clo = T.block_end, chi = T.block_first;          // this chunk's blocks
        if (!chain_par_chunk(v, T, chunks[T.chunk_first + c], lane, 0, pc, clo, chi, &why) || n + pc.total > cap) { give_up = true; break; }
        klo = min(klo, clo); khi = max(khi, chi);
        // ---- list: the same steps again (the bytes are in the cache now), every record with its virtual offsets: the
        //      places only go up, so the block a place lies in is found by stepping on from the one before ----
        if (pc.cnt > 0) {
            int64_t p = pc.s;
            WalkRec* o = mine + n + pc.first;
            VoffCursor vc;
            vc.start(v, clo, chi, p);
            uint64_t at = vc.at(v, p);
            for (int q = 0; q < pc.cnt; ++q) {
                const int64_t nxt = p + 4 + (int64_t)(int32_t)g_u32(out, p);
                const uint64_t after = vc.at(v, nxt);
                o[q] = WalkRec{p, at, after};
                p = nxt;
                at = after;
            }
        }
        n += pc.total;
    }
    if (lane == 0) chained[t] = give_up ? WalkChained{WALK_OK, 0, 0, 0, 0, why} : WalkChained{WALK_OK, (int32_t)n, 1, klo, khi, 0};
}

__global__ void __launch_bounds__(LANES) walk_chain_kernel(WalkView v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks,
                                                           const int64_t* rec_base, WalkRec* recs, WalkChained* chained) {
    __shared__ __attribute__((aligned(16))) uint8_t window[WALK_WINDOW];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (walk_uniform((uint32_t)chained[t].mode) == 1u) return;      // (walk_chain_par_kernel listed this region)
    const tredgpu_walk_task T = tasks[t];
    WalkLds S;
    S.cap = 0; S.mask = 0; S.table = nullptr;
    S.window = (lds_u8*)window;
    WalkReader rd;
    rd.out = v.out; rd.out_end = v.out_end; rd.S = &S; rd.base = (int64_t)1 << 60; rd.lane = lane; rd.ahead_base = (int64_t)1 << 60;
    WalkRec* mine = recs + rec_base[t];
    const int64_t cap = rec_base[t + 1] - rec_base[t];
    int64_t n = 0;
    int status = T.n_chunks < 0 ? WALK_NOT_PLANNED : WALK_OK;
    for (int c = 0; status == WALK_OK && c < T.n_chunks; ++c) {
        const tredgpu_walk_chunk ch = chunks[T.chunk_first + c];
        if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) { status = WALK_NOT_PLANNED; break; }
        WalkCursor cur;
        int rc = cur.enter(v, ch.begin_block);
        if (rc) { status = rc; break; }
        cur.upos = ch.begin_upos;
        bool chunk_done = false;
        while (!chunk_done && status == WALK_OK) {
            int64_t my_a0 = 0;
            uint64_t my_at = 0, my_after = 0;
            const int nb = chain_batch(v, T, ch, cur, rd, S, lane, my_a0, my_at, my_after, chunk_done, status);
            if (n + nb > cap) { status = WALK_TABLE_FULL; break; }
            if (lane < nb) mine[n + lane] = WalkRec{my_a0, my_at, my_after};
            n += nb;
        }
    }
    if (lane == 0) chained[t] = WalkChained{status, (int32_t)n, 0, 0, 0, chained[t].pad};
}

__device__ inline uint32_t g_u8(const uint8_t* out, int64_t at) { return out[at]; }
__device__ inline uint32_t g_u16(const uint8_t* out, int64_t at) { uint16_t x; __builtin_memcpy(&x, out + at, 2); return x; }
__device__ inline uint32_t g_u32(const uint8_t* out, int64_t at) { uint32_t x; __builtin_memcpy(&x, out + at, 4); return x; }

// one lane per record of the call: slot g of the records' pool belongs to the region whose [rec_base[t], rec_base[t+1])
// holds it (binary search), and is record g - rec_base[t] of it -- when that region's chain listed that many
__global__ void __launch_bounds__(256) walk_parse_kernel(WalkView v, int n_tasks, const int64_t* rec_base, const WalkRec* recs,
                                                          const WalkChained* chained, WalkFields* fields) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= rec_base[n_tasks]) return;
    int lo = 0, hi = n_tasks;                              // the last t with rec_base[t] <= g
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rec_base[mid] <= g) lo = mid; else hi = mid;
    }
    const WalkChained C = chained[lo];
    if (C.status != WALK_OK || g - rec_base[lo] >= C.n) return;
    const uint8_t* out = v.out;
    const int64_t a0 = recs[g].a0, r = a0 + 4;
    const int32_t size = (int32_t)g_u32(out, a0);
    WalkFields F;
    F.rtid = (int32_t)g_u32(out, r);
    F.rpos = (int32_t)g_u32(out, r + 4);
    const uint32_t l_name = g_u8(out, r + 8), n_cigar = g_u16(out, r + 12);
    F.flag = (uint16_t)g_u16(out, r + 14);
    const int32_t l_seq = (int32_t)g_u32(out, r + 16);
    F.rend = -1; F.lead = 0; F.trail = 0; F.nlen = 0; F.h = 0;
    F.bad = (l_seq < 0 || 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)l_seq + 1) / 2 > (int64_t)size) ? 1u : 0u;
    if (!F.bad) {
        const int64_t cig = r + 32 + l_name;
        if (!(F.flag & 0x4) && n_cigar > 0) {
            int64_t e = F.rpos;
            for (uint32_t q = 0; q < n_cigar; ++q) {
                const uint32_t op = g_u32(out, cig + 4 * q);
                if ((0x18Du >> (op & 15)) & 1) e += op >> 4;   // M D N = X consume the reference
            }
            F.rend = (int32_t)e;
        }
        for (uint32_t q = 0; q < n_cigar; ++q) {               // query_alignment_start: leading soft clips
            const uint32_t op = g_u32(out, cig + 4 * q);
            if ((op & 15) == 4) F.lead += (int32_t)(op >> 4);
            else if ((op & 15) == 5) continue;
            else break;
        }
        for (int q = (int)n_cigar - 1; q >= 0; --q) {          // query_length - query_alignment_end
            const uint32_t op = g_u32(out, cig + 4 * q);
            if ((op & 15) == 4) F.trail += (int32_t)(op >> 4);
            else if ((op & 15) == 5) continue;
            else break;
        }
        const uint32_t nlen = l_name > 0 ? l_name - 1 : 0;
        F.nlen = (uint16_t)nlen;
        uint32_t h = 2166136261u ^ nlen;                       // (any hash will do: names are compared byte for byte at the end)
        for (uint32_t q = 0; q < nlen; ++q) h = (h ^ g_u8(out, r + 32 + q)) * 16777619u;
        h ^= h >> 15;
        h *= 0x2C1B3C6Du;
        h ^= h >> 12;
        F.h = h;
    }
    fields[g] = F;
}

// resolve: PairTable::add and PairTable::finish for the listed records of a region -- a workgroup of PW_WAVES wavefronts per
// region, every step by all lanes at once.  (Round 5's first version took the records through the table one after the
// other on the scalar unit of ONE wavefront: ~80 instructions and an LDS round trip per record, 1.3 of the kernel's 1.5 ms;
// then one wavefront with the passes below: 0.63 ms, all of it waiting for its own loads.)  What PairTable::add computes is,
// per name, its first and its second record in file order, and the names in order of their first record; none of that
// needs the records one at a time.  A slot of the table is 64 bits: hash bits 6..31 | first record | second record
// (19 bits each, all ones = none), found by compare-and-swap on the hash (linear probing, insert only):
//   pass 1  every record of a pair-forming read puts (hash | its number | none) into its name's slot with an atomic MIN:
//           the slot ends up with the name's FIRST record; the window's records are counted on the way;
//   pass 2  the records that find their own number there are the first ones: their ballot per batch of 64 is kept, a prefix
//           sum over the batches turns it into the pair's index (order of first appearance, as the serial table had it);
//           the others put their number into the low bits, again with an atomic MIN: the name's SECOND record;
//   pass 3  first and second records write their side of the pair; a third, fourth ... record only has to bear the pair's
//           name (a hash is not a name: what it does not prove is checked byte for byte, here and in finish);
//   finish  as before -- names equal? orientation, length, which list -- with the lists' places from a prefix sum over the
//           batches of pairs.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
constexpr int PW_WAVES = 8, PW_THREADS = PW_WAVES * LANES;
constexpr int RESOLVE_MAX_RECORDS = 32768;         // (the per-batch arrays below; 2 x WALK_PAIR_CAP names' worth of records)
constexpr int RESOLVE_MAX_BATCHES = RESOLVE_MAX_RECORDS / LANES, PAIR_MAX_BATCHES = WALK_PAIR_CAP / LANES;
constexpr uint32_t REC_NONE = 0x7FFFFu;
constexpr int REC_BITS = 19;

__device__ inline bool walk_todo(const tredgpu_walk_task& T, const WalkFields& F, bool mine) {
    const bool off_region = F.rtid != T.tid || F.rpos >= T.end;             // (rtid < T.tid: the chain ends at the others)
    const int64_t e = (F.rend < 0 || F.rend <= F.rpos) ? (int64_t)F.rpos + 1 : (int64_t)F.rend;
    return mine && !off_region && e > T.start && (F.flag & 0x1) && !(F.flag & 0x4) && !(F.flag & 0x400);
}
// the slot of a name that is in the table
__device__ inline uint32_t walk_slot_of(const lds_u64* tab, uint32_t mask, uint32_t h, uint64_t* cur) {
    uint32_t slot = h & mask;
    for (uint32_t tries = 0; tries <= mask; ++tries, slot = (slot + 1) & mask) {
        *cur = tab[slot];
        if ((uint32_t)(*cur >> (2 * REC_BITS)) == h >> 6) break;
    }
    return slot;
}

__device__ inline bool walk_same_name(const uint8_t* out, int64_t a, int64_t b, uint32_t len) {
    uint64_t diff = 0;
    uint32_t q = 0;
    for (; q + 8 <= len; q += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, out + a + q, 8);
        __builtin_memcpy(&y, out + b + q, 8);
        diff |= x ^ y;
    }
    for (; q < len; ++q) diff |= (uint64_t)(out[a + q] ^ out[b + q]);
    return diff == 0;
}

struct PairShared {
    uint64_t first_bits[RESOLVE_MAX_BATCHES];      // per batch of records: which of them are a name's first record
    int32_t first_base[RESOLVE_MAX_BATCHES];       // first records per batch, then (exclusive prefix) the first pair index of the batch
    int32_t cg[PAIR_MAX_BATCHES], ct[PAIR_MAX_BATCHES];   // per batch of pairs: lengths for the global / the target list, then their places
    uint64_t wvbeg[PW_WAVES], wvend[PW_WAVES];     // per wavefront: the window's records it saw (first / last batch, offsets, count)
    int32_t wfirst[PW_WAVES], wlast[PW_WAVES], wn[PW_WAVES];
    int64_t firsts[2];
    int32_t status, inserted, np, ng, nt, clash, no_end;
};

// exclusive prefix sum of a[0 .. m) in place, by one wavefront; returns the total
__device__ inline int walk_scan_lds(int32_t* a, int m, int lane) {
    int carry = 0;
    for (int base = 0; base < m; base += LANES) {
        const int val = base + lane < m ? a[base + lane] : 0;
        const int incl = wave_incl_scan(val);
        if (base + lane < m) a[base + lane] = carry + incl - val;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    return carry;
}

__global__ void __launch_bounds__(PW_THREADS) pair_walk_kernel(WalkView v, const tredgpu_walk_task* tasks, const int64_t* rec_base,
                                                               const WalkRec* recs_all, const WalkFields* fields_all, const WalkChained* chained,
                                                               tredgpu_walk_result* results, WalkPair* pairs_all, int32_t* gpool,
                                                               int64_t cap_g, int32_t* tpool, int64_t cap_t, unsigned long long* counters,
                                                               int table_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t walk_lds[];
    __shared__ PairShared sh;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & (LANES - 1), w = tid / LANES;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    lds_u64* tab = (lds_u64*)walk_lds;                     // 2 * table_cap slots, open addressing at a load below one half
    const uint32_t mask = 2u * (uint32_t)table_cap - 1;
    for (int k = tid; k < 2 * table_cap; k += PW_THREADS) tab[k] = 0;
    WalkPair* pairs = pairs_all + (size_t)t * WALK_PAIR_CAP;
    const tredgpu_walk_task T = tasks[t];
    const WalkChained C = chained[t];
    const WalkRec* recs = recs_all + rec_base[t];
    const WalkFields* fields = fields_all + rec_base[t];
    const int n = C.n, nb = (n + LANES - 1) / LANES;
    if (tid == 0) {
        sh.status = C.status != WALK_OK ? C.status : (n > RESOLVE_MAX_RECORDS ? (int)WALK_TABLE_FULL : (int)WALK_OK);
        sh.inserted = 0; sh.np = 0; sh.ng = 0; sh.nt = 0; sh.clash = 0; sh.no_end = 0;
    }
    __syncthreads();
    int status = sh.status;
    // ---- pass 1: the window's records; every name's first record ----
    if (status == WALK_OK) {
        int nwin = 0, bfirst = -1, blast = -1;
        uint64_t vbeg = 0, vend = 0;
        // (the wavefront's next batch is loaded while this one goes through the table)
        WalkFields Fn = {};
        WalkRec men = {};
        if (w * LANES + lane < n) { Fn = fields[w * LANES + lane]; men = recs[w * LANES + lane]; }
        for (int b = w; b < nb; b += PW_WAVES) {
            const int r = b * LANES + lane;
            const bool mine = r < n;
            const WalkFields F = Fn;
            const WalkRec me = men;
            if (r + PW_THREADS < n) { Fn = fields[r + PW_THREADS]; men = recs[r + PW_THREADS]; }
            const int32_t rtid = F.rtid, rpos = F.rpos, rend = F.rend;
            const bool off_region = mine && (rtid != T.tid || rpos >= T.end);
            if (__ballot(mine && !off_region && F.bad != 0) != 0) { if (lane == 0) sh.status = WALK_BAD_RECORD; break; }
            const int64_t e = (rend < 0 || rend <= rpos) ? (int64_t)rpos + 1 : (int64_t)rend;
            const bool keep = mine && !off_region && e > T.start;
            const uint64_t win_mask = __ballot(keep && rpos < T.win_hi && e > T.win_lo);
            if (win_mask) {                                                // records of the scan's own window
                const int wf = __builtin_ctzll(win_mask), wl = 63 - __builtin_clzll(win_mask);
                if (nwin == 0) { vbeg = walk_lane64(me.at, wf); bfirst = b; }
                vend = walk_lane64(me.after, wl); blast = b;
                nwin += __popcll(win_mask);
            }
            bool fresh = false;
            if (walk_todo(T, F, mine)) {
                const uint64_t key = (uint64_t)(F.h >> 6) << (2 * REC_BITS);
                const uint64_t mine64 = key | (uint64_t)(uint32_t)r << REC_BITS | REC_NONE;
                uint32_t slot = F.h & mask;
                for (;; slot = (slot + 1) & mask) {
                    uint64_t cur = tab[slot];
                    if (cur == 0) {
                        unsigned long long expect = 0;
                        if (__hip_atomic_compare_exchange_strong(tab + slot, &expect, (unsigned long long)mine64, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_WORKGROUP)) { fresh = true; break; }
                        cur = expect;
                    }
                    if ((cur >> (2 * REC_BITS)) == (key >> (2 * REC_BITS))) {
                        __hip_atomic_fetch_min(tab + slot, (unsigned long long)mine64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        break;
                    }
                }
            }
            // (2 * cap slots, names counted before the table can fill: the probes above always end)
            const int add = __popcll(__ballot(fresh));
            int seen = 0;
            if (lane == 0) {
                seen = add ? atomicAdd(&sh.inserted, add) + add : sh.inserted;
                if (seen > table_cap) sh.status = WALK_TABLE_FULL;
            }
            if (__builtin_amdgcn_readfirstlane(seen) > table_cap) break;
        }
        if (lane == 0) { sh.wfirst[w] = bfirst; sh.wlast[w] = blast; sh.wn[w] = nwin; sh.wvbeg[w] = vbeg; sh.wvend[w] = vend; }
    }
    __syncthreads();
    status = sh.status;
    // ---- pass 2: which records are first ones; every name's second record ----
    if (status == WALK_OK) {
        for (int b = w; b < nb; b += PW_WAVES) {
            const int r = b * LANES + lane;
            const bool mine = r < n;
            WalkFields F = {};
            if (mine) F = fields[r];
            const bool todo = walk_todo(T, F, mine);
            uint64_t cur = 0;
            uint32_t slot = 0;
            if (todo) slot = walk_slot_of(tab, mask, F.h, &cur);
            const bool is_first = todo && ((uint32_t)(cur >> REC_BITS) & REC_NONE) == (uint32_t)r;
            const uint64_t fm = __ballot(is_first);
            if (lane == 0) { sh.first_bits[b] = fm; sh.first_base[b] = __popcll(fm); }
            if (todo && !is_first)
                __hip_atomic_fetch_min(tab + slot, (unsigned long long)((cur & ~(uint64_t)REC_NONE) | (uint32_t)r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (status == WALK_OK && w == 0) {
        const int total = walk_scan_lds(sh.first_base, nb, lane);
        if (lane == 0) sh.np = total;
    }
    __syncthreads();
    const int np = sh.np;
    // ---- pass 3: both sides of every pair; further records under a pair's hash must bear its name ----
    if (status == WALK_OK) {
        bool clash = false;
        for (int b = w; b < nb; b += PW_WAVES) {
            const int r = b * LANES + lane;
            const bool mine = r < n;
            WalkFields F = {};
            if (mine) F = fields[r];
            if (walk_todo(T, F, mine)) {
                uint64_t cur;
                walk_slot_of(tab, mask, F.h, &cur);
                const uint32_t first = (uint32_t)(cur >> REC_BITS) & REC_NONE, second = (uint32_t)cur & REC_NONE;
                const int fb = (int)(first / LANES), fl = (int)(first % LANES);
                WalkPair& P = pairs[sh.first_base[fb] + __popcll(sh.first_bits[fb] & (((uint64_t)1 << fl) - 1))];
                const int64_t name_at = recs[r].a0 + 36;
                if ((uint32_t)r == first) {
                    P.name_at = name_at; P.name_len = F.nlen;
                    P.a_pos = F.rpos; P.a_lead = F.lead; P.a_rev = (F.flag & 0x10) ? 1 : 0; P.complete = second != REC_NONE ? 1 : 0;
                } else if ((uint32_t)r == second) {
                    P.name2_at = name_at;
                    P.b_end = F.rend; P.b_trail = F.trail; P.b_rev = (F.flag & 0x10) ? 1 : 0;
                } else clash |= fields[first].nlen != F.nlen || !walk_same_name(v.out, recs[first].a0 + 36, name_at, F.nlen);
            }
        }
        if (__ballot(clash) != 0 && lane == 0) sh.clash = 1;
    }
    __syncthreads();                                       // the pair entries are visible to the workgroup
    // ---- PairTable::finish, 64 pairs at a time: are the names under one hash equal? which list does the pair go to? ----
    auto classify = [&](int q, int32_t& len32, bool& clash, bool& no_end) {
        int cls = 0;                                       // 1 global, 2 target
        if (q < np && pairs[q].complete) {
            const WalkPair P = pairs[q];
            clash |= !walk_same_name(v.out, P.name_at, P.name2_at, P.name_len);
            if (!P.a_rev && P.b_rev) {                     // mapped in +, - orientation
                if (P.b_end < 0) no_end = true;            // (the reference dies here: the host reports it)
                const int64_t tlen = ((int64_t)P.b_end + P.b_trail) - ((int64_t)P.a_pos - P.a_lead);
                if (tlen < T.span) { cls = (P.a_pos < T.tstart && P.b_end > T.tend) ? 2 : 1; len32 = (int32_t)tlen; }
            }
        }
        return cls;
    };
    const int npb = (np + LANES - 1) / LANES;
    if (status == WALK_OK) {
        bool clash = false, no_end = false;
        for (int pb = w; pb < npb; pb += PW_WAVES) {
            int32_t len32 = 0;
            const int cls = classify(pb * LANES + lane, len32, clash, no_end);
            const int g = __popcll(__ballot(cls == 1)), tt = __popcll(__ballot(cls == 2));
            if (lane == 0) { sh.cg[pb] = g; sh.ct[pb] = tt; }
        }
        if (__ballot(clash) != 0 && lane == 0) sh.clash = 1;
        if (__ballot(no_end) != 0 && lane == 0) sh.no_end = 1;
    }
    __syncthreads();
    if (status == WALK_OK) status = sh.clash ? (int)WALK_TAG_CLASH : (sh.no_end ? (int)WALK_NO_END : (int)WALK_OK);
    if (status == WALK_OK && w == 0) {
        const int ng = walk_scan_lds(sh.cg, npb, lane), nt = walk_scan_lds(sh.ct, npb, lane);
        if (lane == 0) {
            sh.ng = ng; sh.nt = nt;
            sh.firsts[0] = (int64_t)atomicAdd(&counters[0], (unsigned long long)ng);
            sh.firsts[1] = (int64_t)atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
    __syncthreads();
    tredgpu_walk_result R = {};
    if (status == WALK_OK) {
        const int64_t gf = sh.firsts[0], tf = sh.firsts[1];
        const int ng = sh.ng, nt = sh.nt;
        if (gf + ng > cap_g || tf + nt > cap_t) status = WALK_POOL_FULL;
        else {
            for (int pb = w; pb < npb; pb += PW_WAVES) {
                int32_t len32 = 0;
                bool c1 = false, c2 = false;
                const int cls = classify(pb * LANES + lane, len32, c1, c2);
                const uint64_t mg = __ballot(cls == 1), mt = __ballot(cls == 2);
                if (cls == 1) gpool[gf + sh.cg[pb] + __popcll(mg & below)] = len32;
                if (cls == 2) tpool[tf + sh.ct[pb] + __popcll(mt & below)] = len32;
            }
            R.n_global = ng; R.n_target = nt; R.global_first = gf; R.target_first = tf;
            // the window's records: the count of all wavefronts, the first one's and the last one's offsets
            int nwin = 0, bf = 1 << 30, bl = -1;
            for (int k = 0; k < PW_WAVES; ++k) {
                nwin += sh.wn[k];
                if (sh.wn[k] > 0 && sh.wfirst[k] < bf) { bf = sh.wfirst[k]; R.win_vbeg = sh.wvbeg[k]; }
                if (sh.wn[k] > 0 && sh.wlast[k] > bl) { bl = sh.wlast[k]; R.win_vend = sh.wvend[k]; }
            }
            R.n_window = nwin;
        }
    }
    if (tid == 0) {
        R.status = status;
        if (status != WALK_OK) { R.n_global = R.n_target = R.n_window = 0; R.global_first = R.target_first = 0; R.win_vbeg = R.win_vend = 0; }
        results[t] = R;
    }
}

// ---- the same walk over an alternative locus: the records whose mate lies in the locus' window -------------------------------
// BamParser.parse's mate rescue (tredparse/bam_parser.py:226-243; bamread.cpp scan_impl): per locus ~50 regions of 300 bp
// elsewhere in the genome; a record of such a region counts when its mate maps into the window of the locus.  1 500 tiny
// walks per sample, each from the start of its 16 kb index bin -- on the host they were four fifths of what a scan still
// cost once the pair walks had left it, and their blocks (every region somewhere else in the file) two thirds of what
// still crossed the bus.  Chain and parse as above, no table: a ballot finds the records that count; their virtual offsets
// go into the region's result (at most six: the region is the host's otherwise) and the blocks they lie in are marked
// for the copy back.
constexpr int ALT_MATCH_CAP = 6;
// the block of [lo, hi) that holds byte `addr` of `out` (the last one that begins at or before it: empty blocks hold nothing)
__device__ inline int walk_block_of(const WalkView& v, int lo, int hi, int64_t addr) {
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (v.ooff[mid] <= addr) lo = mid; else hi = mid;
    }
    return lo;
}

// what a record of an alternative locus' region is asked: does it count (its mate lies in the locus' window), is it sound
struct AltRecord { bool bad, hit; int32_t size; };
__device__ inline AltRecord walk_alt_record(const uint8_t* out, const tredgpu_walk_task& T, int64_t a0) {
    AltRecord A;
    const int64_t r = a0 + 4;
    A.size = (int32_t)g_u32(out, a0);
    const int32_t rtid = (int32_t)g_u32(out, r), rpos = (int32_t)g_u32(out, r + 4);
    const uint32_t l_name = g_u8(out, r + 8), n_cigar = g_u16(out, r + 12), flag = g_u16(out, r + 14);
    const int32_t l_seq = (int32_t)g_u32(out, r + 16), mtid = (int32_t)g_u32(out, r + 20), mpos = (int32_t)g_u32(out, r + 24);
    const bool off_region = rtid != T.tid || rpos >= T.end;               // (a contig before the region's: the chain stops at the others)
    A.bad = !off_region && (l_seq < 0 || 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)l_seq + 1) / 2 > (int64_t)A.size);
    A.hit = false;
    if (!A.bad && !off_region && mtid == T.tstart && mpos >= T.win_lo && mpos <= T.win_hi) {
        // (the CIGAR only of the few records whose mate lies in the window: the others cannot count whatever their end)
        int64_t e = (int64_t)rpos + 1;
        if (!(flag & 0x4) && n_cigar > 0) {
            const int64_t cig = r + 32 + l_name;
            int64_t end = rpos;
            for (uint32_t q = 0; q < n_cigar; ++q) {
                const uint32_t op = g_u32(out, cig + 4 * q);
                if ((0x18Du >> (op & 15)) & 1) end += op >> 4;
            }
            if ((int32_t)end > rpos) e = (int32_t)end;
        }
        A.hit = e > T.start;
    }
    return A;
}
// a record that counts: its virtual offset into the region's result (R of the owning lane), its blocks marked for the copy back
__device__ inline void walk_alt_hit(const WalkView& v, int lo, int hi, tredgpu_alt_result& R, int idx, uint64_t at, int64_t a0, int32_t size, uint8_t* need) {
    for (int m = 0; m < ALT_MATCH_CAP; ++m) if (m == idx) R.vbeg[m] = at;
    const int kb = walk_block_of(v, lo, hi, a0), ka = walk_block_of(v, lo, hi, a0 + 3 + (int64_t)size);   // first and last byte
    for (int k = kb; k <= ka; ++k) need[k] = 1;
}

// one chunk, the records one batch after the other (chain_batch: the window in LDS)
__device__ int walk_alt_chunk_serial(const WalkView& v, const tredgpu_walk_task& T, const tredgpu_walk_chunk& ch, WalkReader& rd, WalkLds& S,
                                     tredgpu_alt_result& R, int& found, uint8_t* need, int lane) {
    const uint8_t* out = v.out;
    WalkCursor cur;
    int rc = cur.enter(v, ch.begin_block);
    if (rc) return rc;
    cur.upos = ch.begin_upos;
    bool chunk_done = false;
    while (!chunk_done) {
        int err = WALK_OK;
        int64_t my_a0 = 0;
        uint64_t my_at = 0, my_after = 0;
        const int nb = chain_batch(v, T, ch, cur, rd, S, lane, my_a0, my_at, my_after, chunk_done, err);
        if (err != WALK_OK) chunk_done = true;
        AltRecord A = {false, false, 0};
        if (lane < nb) A = walk_alt_record(out, T, my_a0);
        const uint64_t bad_mask = __ballot(A.bad);
        const int limit = bad_mask ? __builtin_ctzll(bad_mask) : 64;
        uint64_t hits = __ballot(A.hit && lane < limit);
        while (hits) {
            const int j = __builtin_ctzll(hits);
            hits &= hits - 1;
            if (found >= ALT_MATCH_CAP) return WALK_POOL_FULL;
            if (lane == j) walk_alt_hit(v, T.block_first, T.block_end, R, found, my_at, my_a0, A.size, need);
            ++found;
        }
        if (bad_mask) return WALK_BAD_RECORD;
        if (err != WALK_OK) return err;
    }
    return WALK_OK;
}

// one chunk, every lane the records chain_par_chunk gave it
__device__ int walk_alt_chunk_par(const WalkView& v, const tredgpu_walk_task& T, const ParChunk& pc, int klo, int khi,
                                  tredgpu_alt_result& R, int& found, uint8_t* need, int lane) {
    const uint8_t* out = v.out;
    int nh = 0;
    bool bad = false;
    int64_t hit_a0[ALT_MATCH_CAP];
    int32_t hit_size[ALT_MATCH_CAP];
    int64_t p = pc.s;
    for (int q = 0; q < pc.cnt; ++q) {
        const AltRecord A = walk_alt_record(out, T, p);
        bad |= A.bad;
        if (A.hit) {
            for (int m = 0; m < ALT_MATCH_CAP; ++m) if (m == nh) { hit_a0[m] = p; hit_size[m] = A.size; }
            ++nh;
        }
        p += 4 + (int64_t)A.size;
    }
    if (__ballot(bad) != 0) return WALK_BAD_RECORD;
    const int incl = wave_incl_scan(nh);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (found + total > ALT_MATCH_CAP) return WALK_POOL_FULL;
    for (int m = 0; m < ALT_MATCH_CAP; ++m)
        if (m < nh) walk_alt_hit(v, klo, khi, R, found + incl - nh + m, walk_voffset(v, klo, khi, hit_a0[m]), hit_a0[m], hit_size[m], need);
    found += total;
    return WALK_OK;
}

constexpr int64_t ALT_PAR_MIN_COMP = 8192;        // compressed bytes of a chunk from which the lanes share it (~150 records)
__device__ int walk_alt_records(const WalkView& v, const tredgpu_walk_task& T, const tredgpu_walk_chunk* chunks, WalkReader& rd, WalkLds& S,
                                tredgpu_alt_result& R, uint8_t* need, int lane) {
    if (T.n_chunks < 0) return WALK_NOT_PLANNED;
    int found = 0;
    for (int c = 0; c < T.n_chunks; ++c) {
        const tredgpu_walk_chunk ch = chunks[T.chunk_first + c];
        if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) return WALK_NOT_PLANNED;
        // A region's walk starts where its 16 kb bin starts: a few records as a rule, thousands when the bin lies in a covered
        // stretch -- those few regions were what a launch waited for (1.2 of its 1.4 ms).
        bool done = false;
        if ((int64_t)(ch.end_voffset >> 16) - walk_uniform64(v.bcoff[ch.begin_block]) >= ALT_PAR_MIN_COMP) {
            ParChunk pc;
            int klo = T.block_end, khi = T.block_first;
            if (chain_par_chunk(v, T, ch, lane, 0, pc, klo, khi)) {
                const int rc = walk_alt_chunk_par(v, T, pc, klo, khi, R, found, need, lane);
                if (rc) return rc;
                done = true;
            }
        }
        if (!done) {
            const int rc = walk_alt_chunk_serial(v, T, ch, rd, S, R, found, need, lane);
            if (rc) return rc;
        }
    }
    R.n = found;
    return WALK_OK;
}

__global__ void __launch_bounds__(LANES) alt_walk_kernel(WalkView v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks,
                                                         tredgpu_alt_result* results, uint8_t* need) {
    extern __shared__ __attribute__((aligned(16))) uint8_t walk_lds[];
    const int t = blockIdx.x, lane = threadIdx.x;
    WalkLds S;
    S.cap = 0; S.mask = 0; S.table = nullptr;
    S.window = (lds_u8*)walk_lds;
    const tredgpu_walk_task T = tasks[t];
    WalkReader rd;
    rd.out = v.out; rd.out_end = v.out_end; rd.S = &S; rd.base = (int64_t)1 << 60; rd.lane = lane; rd.ahead_base = (int64_t)1 << 60;
    // (every lane holds the result; the lane that owns a record writes that record's offset into ITS copy: gather them)
    tredgpu_alt_result R = {};
    const int status = walk_alt_records(v, T, chunks, rd, S, R, need, lane);
    tredgpu_alt_result out = {};
    out.status = status;
    if (status == WALK_OK) {
        out.n = R.n;
        for (int m = 0; m < ALT_MATCH_CAP; ++m) {
            // the owner's copy is the only non-zero one
            uint64_t x = R.vbeg[m];
            for (int d = 32; d >= 1; d >>= 1) x |= (uint64_t)__shfl_xor((unsigned long long)x, d, 64);
            out.vbeg[m] = m < R.n ? x : 0;
        }
    }
    if (lane == 0) results[t] = out;
}

// ---- the fetch as a kernel ---------------------------------------------------------------------------------------------
// The blocks the host wants, copied from the decoder's output straight into pinned host memory by the GPU's own stores (the
// pinned buffer is mapped into the device's address space): ONE launch and one wait per call where the copy engines
// were handed ~660 copies of ~160 KB (two thirds of a call's time went there once three driver processes shared them).
// piece p: `len` bytes from out + src to host + dst; 256 lanes, 16 bytes each per step.  dst is any address: the bytes up to
// its next 16-byte boundary go one by one, the stores behind them are aligned (the loads never had to be) -- the runs lie in
// the dense buffer without padding, so that every block's length can be read off dense_off.
struct FetchPiece { int64_t src, dst; int32_t len, pad; };
__global__ void __launch_bounds__(256) fetch_gather_kernel(const uint8_t* __restrict__ out, uint8_t* __restrict__ host, const FetchPiece* pieces) {
    const FetchPiece P = pieces[blockIdx.x];
    const uint8_t* s = out + P.src;
    uint8_t* d = host + P.dst;
    const int head = min(P.len, (int)((16 - (P.dst & 15)) & 15));
    if ((int)threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
    const int whole = (P.len - head) & ~15;
    for (int o = head + threadIdx.x * 16; o < head + whole; o += 256 * 16) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 __attribute__((aligned(1))) u32x4_any;
        *reinterpret_cast<u32x4*>(d + o) = *reinterpret_cast<const u32x4_any*>(s + o);
    }
    const int tail = head + whole + (int)threadIdx.x;
    if (tail < P.len) d[tail] = s[tail];
}

}  // namespace

// ---- C ABI (include/tredgpu.h) ------------------------------------------------------------------------------------
// A call is cut into slices of blocks that alternate between two streams: the copy-in and the decoding of slice k + 1
// run beside the copy-out of slice k (the copy-out is the long pole: four bytes leave for every byte that arrives).
constexpr int MAX_SLICES = 8;
constexpr int SLICE_BLOCKS = 4096;      // >= 4 096 wavefronts per launch: 16 per CU, and two launches run side by side

struct tredgpu_inflater {
    int device = 0;
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};   // waited for asleep (polled): see tredgpu_inflate_blocks
    hipEvent_t t0[2] = {}, t1[2] = {}, k0[MAX_SLICES] = {}, k1[MAX_SLICES] = {};   // timing (tredgpu_inflater_timing)
    int last_slices = 0, last_streams = 0;
    uint8_t *h_comp = nullptr, *h_out = nullptr;      // pinned staging the caller fills / reads in place
    bool host_out = true;                             // false (tredgpu_inflater_host_out): no pinned room for the whole output --
                                                      // the blocks the host wants come through tredgpu_inflater_fetch_dense
    uint8_t* h_dense = nullptr; size_t cap_dense = 0; // pinned: the fetched blocks, one after the other
    uint8_t *h_pieces = nullptr, *d_pieces = nullptr; size_t cap_pieces = 0;   // the fetch kernel's copy table (FetchPiece)
    int64_t *h_off = nullptr;                         // pinned: comp_off[n+1] then out_off[n+1]
    int32_t* h_status = nullptr;                      // pinned: status[n] then crc[n]
    size_t cap_comp = 0, cap_out = 0, cap_blocks = 0;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    int64_t* d_off = nullptr;
    int32_t* d_status = nullptr;
    // the pair walk (tredgpu_inflate_walk): a stream of its own, the file's view of the blocks, tasks, per-task tables, pools
    hipStream_t wstream = nullptr, astream = nullptr;          // (astream: the alternative loci's walks, beside the pair walks)
    hipEvent_t adone = nullptr;
    hipEvent_t wdone = nullptr, w0 = nullptr, w1 = nullptr, decoded[2] = {nullptr, nullptr};
    bool walk_timed = false, big_lds_allowed = false;
    int walk_table_cap = 0;
    size_t last_walk_tasks = 0;                       // regions of the last walk call (tredgpu_inflater_walk_serial_regions)
    uint8_t* d_wblk = nullptr;  size_t cap_wblk = 0;        // bcoff[n] int64, then bclen[n] int32, then xcrc[n] uint32
    uint8_t* h_wblk = nullptr;                               // pinned, same layout
    uint8_t* d_wtask = nullptr; uint8_t* h_wtask = nullptr; size_t cap_wtask = 0;   // tasks then chunks
    uint8_t* d_wres = nullptr;  uint8_t* h_wres = nullptr;  size_t cap_wres = 0;    // results then the two counters
    WalkPair* d_wpairs = nullptr; size_t cap_wscratch = 0;   // in tasks
    WalkRec* d_wrecs = nullptr; WalkFields* d_wfields = nullptr; size_t cap_wrecs = 0;          // in records: the chain's list, the parsed fields
    WalkChained* d_wchained = nullptr; size_t cap_wchained = 0;
    uint8_t* d_atask = nullptr; uint8_t* h_atask = nullptr; size_t cap_atask = 0;   // the alternative loci's tasks then chunks
    uint8_t* d_ares = nullptr;  uint8_t* h_ares = nullptr;  size_t cap_ares = 0;    // their results, then the blocks' need flags
    int32_t *d_gpool = nullptr, *d_tpool = nullptr, *h_gpool = nullptr, *h_tpool = nullptr;
    size_t cap_gpool = 0, cap_tpool = 0;          // (device pools: the call's bound)
    size_t cap_hgpool = 0, cap_htpool = 0;        // (pinned host pools: what the walks really produced, an eighth more)
    std::string err;
};

namespace {
thread_local std::string g_inflate_error;

// TREDGPU_TRACE=1: host-side timestamps of a call's phases on stderr (milliseconds since the call began)
struct CallTrace {
    bool on; const char* what; double t0; std::string line;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    explicit CallTrace(const char* w) : on(getenv("TREDGPU_TRACE") != nullptr), what(w), t0(on ? now() : 0) {}
    void mark(const char* k) { if (on) { char b[64]; snprintf(b, sizeof b, " %s=%.2f", k, now() - t0); line += b; } }
    ~CallTrace() { if (on) fprintf(stderr, "[tredgpu %d] %s:%s\n", (int)getpid(), what, line.c_str()); }
};


int ifail(tredgpu_inflater* f, int code, const char* what, hipError_t e = hipSuccess) {
    std::string m = what;
    if (e != hipSuccess) { m += ": "; m += hipGetErrorString(e); }
    if (f) f->err = m; else g_inflate_error = m;
    return code;
}

#define ICHK(f, expr)                                             \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) return ifail((f), -10, #expr, e_);  \
    } while (0)

void release(tredgpu_inflater* f) {
    if (f->h_comp) (void)hipHostFree(f->h_comp);
    if (f->h_out) (void)hipHostFree(f->h_out);
    if (f->h_dense) (void)hipHostFree(f->h_dense);
    f->h_dense = nullptr; f->cap_dense = 0;
    if (f->h_pieces) (void)hipHostFree(f->h_pieces);
    if (f->d_pieces) (void)hipFree(f->d_pieces);
    f->h_pieces = f->d_pieces = nullptr; f->cap_pieces = 0;
    if (f->h_off) (void)hipHostFree(f->h_off);
    if (f->h_status) (void)hipHostFree(f->h_status);
    for (void* p : {(void*)f->d_comp, (void*)f->d_out, (void*)f->d_off, (void*)f->d_status})
        if (p) (void)hipFree(p);
    f->h_comp = f->h_out = nullptr; f->h_off = nullptr; f->h_status = nullptr;
    f->d_comp = f->d_out = nullptr; f->d_off = nullptr; f->d_status = nullptr;
    f->cap_comp = f->cap_out = f->cap_blocks = 0;
}

void release_walk(tredgpu_inflater* f) {
    for (void* p : {(void*)f->h_wblk, (void*)f->h_wtask, (void*)f->h_wres, (void*)f->h_gpool, (void*)f->h_tpool, (void*)f->h_atask, (void*)f->h_ares})
        if (p) (void)hipHostFree(p);
    for (void* p : {(void*)f->d_wblk, (void*)f->d_wtask, (void*)f->d_wres, (void*)f->d_wpairs, (void*)f->d_gpool, (void*)f->d_tpool, (void*)f->d_atask, (void*)f->d_ares,
                    (void*)f->d_wrecs, (void*)f->d_wfields, (void*)f->d_wchained})
        if (p) (void)hipFree(p);
    f->h_wblk = f->h_wtask = f->h_wres = f->h_atask = f->h_ares = nullptr; f->h_gpool = f->h_tpool = nullptr;
    f->d_wblk = f->d_wtask = f->d_wres = f->d_atask = f->d_ares = nullptr; f->d_wpairs = nullptr; f->d_gpool = f->d_tpool = nullptr;
    f->d_wrecs = nullptr; f->d_wfields = nullptr; f->d_wchained = nullptr; f->cap_wrecs = f->cap_wchained = 0;
    f->cap_wblk = f->cap_wtask = f->cap_wres = f->cap_wscratch = f->cap_gpool = f->cap_tpool = f->cap_atask = f->cap_ares = 0;
    f->cap_hgpool = f->cap_htpool = 0;
}

// grow-only pairs of pinned host / device buffers for the walk's small arrays
int grow_pair(tredgpu_inflater* f, uint8_t** host, uint8_t** dev, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    const size_t c = std::max(need, *cap + *cap / 2);
    if (*host) (void)hipHostFree(*host);
    if (*dev) (void)hipFree(*dev);
    *host = nullptr; *dev = nullptr; *cap = 0;
    ICHK(f, hipHostMalloc((void**)host, c, hipHostMallocDefault));
    ICHK(f, hipMalloc((void**)dev, c));
    *cap = c;
    return 0;
}

void destroy_handles(tredgpu_inflater* f) {
    for (hipEvent_t e : {f->done[0], f->done[1], f->t0[0], f->t0[1], f->t1[0], f->t1[1]}) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < MAX_SLICES; ++k) { if (f->k0[k]) (void)hipEventDestroy(f->k0[k]); if (f->k1[k]) (void)hipEventDestroy(f->k1[k]); }
    for (hipEvent_t e : {f->wdone, f->w0, f->w1, f->decoded[0], f->decoded[1]}) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : f->stream) if (st) (void)hipStreamDestroy(st);
    if (f->wstream) (void)hipStreamDestroy(f->wstream);
    if (f->astream) (void)hipStreamDestroy(f->astream);
    if (f->adone) (void)hipEventDestroy(f->adone);
}
}  // namespace

extern "C" {

int tredgpu_inflater_create(int device_id, tredgpu_inflater** out) {
    if (!out) return ifail(nullptr, -2, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return ifail(nullptr, -3, "no HIP device available; libtredgpu has no CPU fallback", e);
    if (device_id < 0 || device_id >= n) return ifail(nullptr, -2, "device out of range");
    tredgpu_inflater* f = new tredgpu_inflater();
    f->device = device_id;
    // the lowest stream priority: a genotyping launch of the same or another driver process should not queue up behind
    // several of these
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    e = hipSetDevice(device_id);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipStreamCreateWithPriority(&f->stream[k], hipStreamNonBlocking, lo_prio);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->done[k], hipEventBlockingSync | hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&f->t0[k]);
        if (e == hipSuccess) e = hipEventCreate(&f->t1[k]);
    }
    for (int k = 0; k < MAX_SLICES && e == hipSuccess; ++k) {
        e = hipEventCreate(&f->k0[k]);
        if (e == hipSuccess) e = hipEventCreate(&f->k1[k]);
    }
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&f->wstream, hipStreamNonBlocking, lo_prio);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&f->astream, hipStreamNonBlocking, lo_prio);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->adone, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->wdone, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&f->w0);
    if (e == hipSuccess) e = hipEventCreate(&f->w1);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&f->decoded[k], hipEventDisableTiming);
    if (e != hipSuccess) {
        destroy_handles(f);
        delete f;
        return ifail(nullptr, -10, "stream / event creation", e);
    }
    *out = f;
    return 0;
}

void tredgpu_inflater_destroy(tredgpu_inflater* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (hipStream_t st : f->stream) (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(f->wstream);
    if (f->astream) (void)hipStreamSynchronize(f->astream);
    release(f);
    release_walk(f);
    destroy_handles(f);
    delete f;
}

const char* tredgpu_inflater_last_error(const tredgpu_inflater* f) { return f ? f->err.c_str() : g_inflate_error.c_str(); }

int tredgpu_inflater_reserve(tredgpu_inflater* f, int64_t comp_bytes, int64_t out_bytes, int32_t n_blocks, uint8_t** comp_host,
                             uint8_t** out_host, int64_t** comp_off_host, int64_t** out_off_host) {
    if (!f) return -2;
    if (comp_bytes < 0 || out_bytes < 0 || n_blocks < 0 || !comp_host || !out_host || !comp_off_host || !out_off_host)
        return ifail(f, -2, "bad arguments");
    ICHK(f, hipSetDevice(f->device));
    const size_t need_c = (size_t)comp_bytes + 64, need_o = (size_t)out_bytes + 64, need_b = (size_t)n_blocks + 1;
    if (need_c > f->cap_comp || need_o > f->cap_out || need_b > f->cap_blocks) {
        for (hipStream_t st : f->stream) ICHK(f, hipStreamSynchronize(st));
        // (page-locked staging grows by an eighth past the largest call seen: a chunk's size varies by a few per cent, and every
        //  spare byte here is pinned three times per driver process)
        const size_t cc = std::max(need_c, f->cap_comp + f->cap_comp / 8), co = std::max(need_o, f->cap_out + f->cap_out / 8),
                     cb = std::max(need_b, f->cap_blocks + f->cap_blocks / 2);
        release(f);
        ICHK(f, hipHostMalloc((void**)&f->h_comp, cc, hipHostMallocDefault));
        if (f->host_out) ICHK(f, hipHostMalloc((void**)&f->h_out, co, hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_off, 2 * cb * sizeof(int64_t), hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_status, 2 * cb * sizeof(int32_t), hipHostMallocDefault));
        ICHK(f, hipMalloc((void**)&f->d_comp, cc));
        ICHK(f, hipMalloc((void**)&f->d_out, co));
        ICHK(f, hipMalloc((void**)&f->d_off, 2 * cb * sizeof(int64_t)));
        ICHK(f, hipMalloc((void**)&f->d_status, 2 * cb * sizeof(int32_t)));
        f->cap_comp = cc; f->cap_out = co; f->cap_blocks = cb;
    }
    *comp_host = f->h_comp;
    *out_host = f->h_out;
    *comp_off_host = f->h_off;
    *out_off_host = f->h_off + f->cap_blocks;
    return 0;
}

}  // extern "C"

namespace {
int wait_asleep(tredgpu_inflater* f, hipEvent_t ev) {
    // hipEventSynchronize spins even on a hipEventBlockingSync event here (measured: CPU time = wall time, and calls of
    // other threads on other streams queue up behind the spinning one); the host threads that wait are the ones whose
    // cores the path is short of
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return ifail(f, -10, "hipEventQuery", q);
        usleep(100);
    }
}

// copy in, decode (in slices on two streams); copy_out: every slice's blocks go back as soon as they are decoded.
// w != nullptr: the pair walk follows the last slice on a stream of its own, and its results come back.
int run_inflate(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc, bool copy_out, tredgpu_walk_args* w) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || (n_blocks > 0 && !status)) return ifail(f, -2, "bad arguments (reserve first)");
    if (n_blocks == 0) return 0;
    if (copy_out && !f->h_out) return ifail(f, -2, "this inflater keeps no host copy of the output (tredgpu_inflater_host_out): walk and fetch");
    CallTrace tr(w ? "inflate_walk" : "inflate");
    const int64_t* coff = f->h_off;
    const int64_t* ooff = f->h_off + f->cap_blocks;
    for (int32_t k = 0; k < n_blocks; ++k) {
        if (coff[k] < 0 || (coff[k] & 3) != 0 || coff[k + 1] < coff[k] || ooff[k] < 0 || ooff[k + 1] < ooff[k] ||
            ooff[k + 1] - ooff[k] > 65536)
            return ifail(f, -2, "block offsets: payloads start on 4-byte boundaries, ascend, and inflate to at most 64 KiB each");
    }
    if ((size_t)coff[n_blocks] + 64 > f->cap_comp || (size_t)ooff[n_blocks] + 64 > f->cap_out) return ifail(f, -2, "offsets beyond the reserved buffers");
    ICHK(f, hipSetDevice(f->device));
    const int slices = std::min(MAX_SLICES, std::max(1, n_blocks / SLICE_BLOCKS));
    const int nstreams = slices > 1 ? 2 : 1;
    int64_t* d_coff = f->d_off;
    int64_t* d_ooff = f->d_off + f->cap_blocks;
    int32_t* d_crc = f->d_status + f->cap_blocks;
    const bool want_crc = crc != nullptr || w != nullptr;
    // ---- the walk's inputs go first, on its own stream (nothing there depends on the decoding yet) ----
    size_t n_tasks = 0, n_chunks = 0, n_alt = 0, n_alt_chunks = 0, total_recs = 0;
    if (w) {
        if (w->n_tasks < 0 || w->n_chunks < 0 || !w->blk_coffset || !w->blk_clen || !w->blk_crc || (w->n_tasks > 0 && (!w->tasks || !w->results)) ||
            (w->n_chunks > 0 && !w->chunks) || w->cap_global < 0 || w->cap_target < 0 || (w->cap_global > 0 && !w->global_pool) ||
            (w->cap_target > 0 && !w->target_pool))
            return ifail(f, -2, "bad walk arguments");
        n_tasks = (size_t)w->n_tasks; n_chunks = (size_t)w->n_chunks;
        for (size_t t = 0; t < n_tasks; ++t) {
            const tredgpu_walk_task& T = w->tasks[t];
            if (T.n_chunks >= 0 && (T.chunk_first < 0 || (size_t)T.chunk_first + (size_t)T.n_chunks > n_chunks || T.block_first < 0 ||
                                    T.block_end > n_blocks || T.block_first > T.block_end))
                return ifail(f, -2, "walk task outside its chunks / blocks");
        }
        for (size_t q = 0; q < n_chunks; ++q)
            if (w->chunks[q].begin_upos < 0 || w->chunks[q].begin_upos > 65536) return ifail(f, -2, "walk chunk starts outside a block");
        n_alt = (size_t)std::max(w->n_alt_tasks, 0);
        n_alt_chunks = (size_t)std::max(w->n_alt_chunks, 0);
        if (w->n_alt_tasks < 0 || w->n_alt_chunks < 0 || (n_alt > 0 && (!w->alt_tasks || !w->alt_results || !w->need)) || (n_alt_chunks > 0 && !w->alt_chunks))
            return ifail(f, -2, "bad walk arguments (alternative loci)");
        for (size_t t = 0; t < n_alt; ++t) {
            const tredgpu_walk_task& T = w->alt_tasks[t];
            if (T.n_chunks >= 0 && (T.chunk_first < 0 || (size_t)T.chunk_first + (size_t)T.n_chunks > n_alt_chunks || T.block_first < 0 ||
                                    T.block_end > n_blocks || T.block_first > T.block_end))
                return ifail(f, -2, "walk task outside its chunks / blocks");
        }
        for (size_t q = 0; q < n_alt_chunks; ++q)
            if (w->alt_chunks[q].begin_upos < 0 || w->alt_chunks[q].begin_upos > 65536) return ifail(f, -2, "walk chunk starts outside a block");
        const size_t nb = (size_t)n_blocks;
        if (n_alt > 0) {
            if (grow_pair(f, &f->h_atask, &f->d_atask, &f->cap_atask, n_alt * sizeof(tredgpu_walk_task) + n_alt_chunks * sizeof(tredgpu_walk_chunk) + 64)) return -10;
            if (grow_pair(f, &f->h_ares, &f->d_ares, &f->cap_ares, n_alt * sizeof(tredgpu_alt_result) + nb + 64)) return -10;
            memcpy(f->h_atask, w->alt_tasks, n_alt * sizeof(tredgpu_walk_task));
            memcpy(f->h_atask + n_alt * sizeof(tredgpu_walk_task), w->alt_chunks, n_alt_chunks * sizeof(tredgpu_walk_chunk));
            ICHK(f, hipMemcpyAsync(f->d_atask, f->h_atask, n_alt * sizeof(tredgpu_walk_task) + n_alt_chunks * sizeof(tredgpu_walk_chunk), hipMemcpyHostToDevice, f->astream));
            ICHK(f, hipMemsetAsync(f->d_ares + n_alt * sizeof(tredgpu_alt_result), 0, nb, f->astream));
        }
        if (grow_pair(f, &f->h_wblk, &f->d_wblk, &f->cap_wblk, nb * 16 + 64)) return -10;
        // tasks | chunks | rec_base[n_tasks + 1] (8-byte entries behind 8-byte-sized structs: aligned)
        const size_t rb_at = n_tasks * sizeof(tredgpu_walk_task) + n_chunks * sizeof(tredgpu_walk_chunk);
        static_assert(sizeof(tredgpu_walk_task) % 8 == 0 && sizeof(tredgpu_walk_chunk) % 8 == 0, "rec_base stays 8-byte aligned");
        if (grow_pair(f, &f->h_wtask, &f->d_wtask, &f->cap_wtask, rb_at + (n_tasks + 1) * sizeof(int64_t) + 64)) return -10;
        // room for every region's record list: the blocks its chunks span (a task's block_first .. block_end is its whole
        // FILE), at no less than 64 bytes per record (36 fixed bytes, a name, 36 bases and their qualities: 100 and more) --
        // a region that has more records than that is the host's (status 4)
        int64_t* rec_base = (int64_t*)(f->h_wtask + rb_at);
        rec_base[0] = 0;
        for (size_t t = 0; t < n_tasks; ++t) {
            const tredgpu_walk_task& T = w->tasks[t];
            int64_t bytes = 0;
            for (int32_t q = 0; q < T.n_chunks; ++q) {
                const tredgpu_walk_chunk& ch = w->chunks[T.chunk_first + q];
                if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) continue;
                const int64_t* lo = w->blk_coffset + ch.begin_block;
                const int64_t* hi = std::upper_bound(lo, w->blk_coffset + T.block_end, (int64_t)(ch.end_voffset >> 16));
                bytes += ooff[ch.begin_block + (hi - lo)] - ooff[ch.begin_block];
            }
            rec_base[t + 1] = rec_base[t] + (T.n_chunks > 0 ? bytes / 64 + 64 : 0);
        }
        total_recs = (size_t)rec_base[n_tasks];
        if (total_recs > f->cap_wrecs) {
            const size_t c = std::max(total_recs, f->cap_wrecs + f->cap_wrecs / 2);
            if (f->d_wrecs) (void)hipFree(f->d_wrecs);
            if (f->d_wfields) (void)hipFree(f->d_wfields);
            f->d_wrecs = nullptr; f->d_wfields = nullptr; f->cap_wrecs = 0;
            ICHK(f, hipMalloc((void**)&f->d_wrecs, c * sizeof(WalkRec)));
            ICHK(f, hipMalloc((void**)&f->d_wfields, c * sizeof(WalkFields)));
            f->cap_wrecs = c;
        }
        if (n_tasks > f->cap_wchained) {
            const size_t c = std::max(n_tasks, f->cap_wchained + f->cap_wchained / 2);
            if (f->d_wchained) (void)hipFree(f->d_wchained);
            f->d_wchained = nullptr; f->cap_wchained = 0;
            ICHK(f, hipMalloc((void**)&f->d_wchained, c * sizeof(WalkChained)));
            f->cap_wchained = c;
        }
        if (grow_pair(f, &f->h_wres, &f->d_wres, &f->cap_wres, n_tasks * sizeof(tredgpu_walk_result) + 64)) return -10;
        if (n_tasks > f->cap_wscratch) {
            const size_t c = std::max(n_tasks, f->cap_wscratch + f->cap_wscratch / 2);
            if (f->d_wpairs) (void)hipFree(f->d_wpairs);
            f->d_wpairs = nullptr; f->cap_wscratch = 0;
            ICHK(f, hipMalloc((void**)&f->d_wpairs, c * WALK_PAIR_CAP * sizeof(WalkPair)));
            f->cap_wscratch = c;
        }
        // (the pools on the device hold the call's BOUND -- a tenth of it is used at 30x --, their pinned host copies are
        //  sized behind the walks, from what was really produced)
        for (int which = 0; which < 2; ++which) {
            int32_t** d = which ? &f->d_tpool : &f->d_gpool;
            size_t* cap = which ? &f->cap_tpool : &f->cap_gpool;
            const size_t need = ((size_t)(which ? w->cap_target : w->cap_global) + 16) * 4;
            if (need > *cap) {
                const size_t c = std::max(need, *cap + *cap / 2);
                if (*d) (void)hipFree(*d);
                *d = nullptr; *cap = 0;
                ICHK(f, hipMalloc((void**)d, c));
                *cap = c;
            }
        }
        memcpy(f->h_wblk, w->blk_coffset, nb * 8);
        memcpy(f->h_wblk + nb * 8, w->blk_clen, nb * 4);
        memcpy(f->h_wblk + nb * 12, w->blk_crc, nb * 4);
        memcpy(f->h_wtask, w->tasks, n_tasks * sizeof(tredgpu_walk_task));
        memcpy(f->h_wtask + n_tasks * sizeof(tredgpu_walk_task), w->chunks, n_chunks * sizeof(tredgpu_walk_chunk));
        ICHK(f, hipMemcpyAsync(f->d_wblk, f->h_wblk, nb * 16, hipMemcpyHostToDevice, f->wstream));
        ICHK(f, hipMemcpyAsync(f->d_wtask, f->h_wtask, rb_at + (n_tasks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->wstream));
        ICHK(f, hipMemsetAsync(f->d_wres + n_tasks * sizeof(tredgpu_walk_result), 0, 16, f->wstream));
    }
    tr.mark("walk_inputs");
    // the two streams never wait for each other: each copies the offsets in for itself (both write the same values)
    for (int s = 0; s < nstreams; ++s) {
        ICHK(f, hipEventRecord(f->t0[s], f->stream[s]));
        ICHK(f, hipMemcpyAsync(d_coff, coff, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream[s]));
        ICHK(f, hipMemcpyAsync(d_ooff, ooff, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream[s]));
    }
    for (int k = 0; k < slices; ++k) {
        hipStream_t st = f->stream[k % nstreams];
        const int32_t b0 = (int32_t)((int64_t)n_blocks * k / slices), b1 = (int32_t)((int64_t)n_blocks * (k + 1) / slices);
        const size_t c_from = (size_t)coff[b0], c_to = std::min(((size_t)coff[b1] + 3) & ~(size_t)3, f->cap_comp);
        ICHK(f, hipMemcpyAsync(f->d_comp + c_from, f->h_comp + c_from, c_to - c_from, hipMemcpyHostToDevice, st));
        ICHK(f, hipEventRecord(f->k0[k], st));
        inflate_kernel<<<b1 - b0, LANES, 0, st>>>((const uint32_t*)f->d_comp, d_coff, f->d_out, d_ooff, b0, f->d_status, want_crc ? (uint32_t*)d_crc : nullptr);
        ICHK(f, hipGetLastError());
        ICHK(f, hipEventRecord(f->k1[k], st));
        if (copy_out) ICHK(f, hipMemcpyAsync(f->h_out + ooff[b0], f->d_out + ooff[b0], (size_t)(ooff[b1] - ooff[b0]), hipMemcpyDeviceToHost, st));
        ICHK(f, hipMemcpyAsync(f->h_status + b0, f->d_status + b0, (size_t)(b1 - b0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (want_crc) ICHK(f, hipMemcpyAsync(f->h_status + f->cap_blocks + b0, d_crc + b0, (size_t)(b1 - b0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    f->last_slices = slices;
    f->last_streams = nstreams;
    f->walk_timed = false;
    if (w) {
        // the walk reads every slice's blocks: its stream waits for the last launch of both decode streams
        for (int s = 0; s < nstreams; ++s) {
            ICHK(f, hipEventRecord(f->decoded[s], f->stream[s]));
            ICHK(f, hipStreamWaitEvent(f->wstream, f->decoded[s], 0));
            ICHK(f, hipStreamWaitEvent(f->astream, f->decoded[s], 0));
        }
        const size_t nb = (size_t)n_blocks;
        WalkView v;
        v.out = f->d_out; v.ooff = d_ooff; v.bstatus = f->d_status; v.bcrc = (const uint32_t*)d_crc;
        v.bcoff = (const int64_t*)f->d_wblk; v.bclen = (const int32_t*)(f->d_wblk + nb * 8); v.xcrc = (const uint32_t*)(f->d_wblk + nb * 12);
        v.out_end = ooff[n_blocks] + 48;                   // (the buffers hold 64 bytes more than reserved)
        ICHK(f, hipEventRecord(f->w0, f->wstream));
        if (n_tasks > 0) {
            // the small table when every region is short: a region of up to 40 blocks (2.6 MB of records, ~8 000 of them)
            // has at most ~4 000 names; one that has more after all is handed back to the host (status 4)
            int table_cap = WALK_PAIR_CAP_SMALL;
            for (size_t t = 0; t < n_tasks && table_cap == WALK_PAIR_CAP_SMALL; ++t) {
                const tredgpu_walk_task& T = w->tasks[t];
                int64_t span = 0;
                for (int32_t q = 0; q < T.n_chunks; ++q) {
                    const tredgpu_walk_chunk& ch = w->chunks[T.chunk_first + q];
                    if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) continue;
                    const int64_t* lo = w->blk_coffset + ch.begin_block;
                    const int64_t* hi = std::upper_bound(lo, w->blk_coffset + T.block_end, (int64_t)(ch.end_voffset >> 16));
                    span += hi - lo;
                }
                if (span > 40) table_cap = WALK_PAIR_CAP;
            }
            if (!f->big_lds_allowed) {                 // (per inflater: each lives on one device, and a process may use several)
                ICHK(f, hipFuncSetAttribute((const void*)pair_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds_bytes(WALK_PAIR_CAP)));
                f->big_lds_allowed = true;
            }
            f->walk_table_cap = table_cap;
            f->last_walk_tasks = n_tasks;
            const tredgpu_walk_task* d_tasks = (const tredgpu_walk_task*)f->d_wtask;
            const tredgpu_walk_chunk* d_chunks = (const tredgpu_walk_chunk*)(f->d_wtask + n_tasks * sizeof(tredgpu_walk_task));
            const int64_t* d_rec_base = (const int64_t*)(f->d_wtask + n_tasks * sizeof(tredgpu_walk_task) + n_chunks * sizeof(tredgpu_walk_chunk));
            walk_chain_par_kernel<<<(unsigned)n_tasks, LANES, 0, f->wstream>>>(v, d_tasks, d_chunks, d_rec_base, f->d_wrecs, f->d_wchained);
            if (getenv("TREDGPU_WALK_SERIAL") != nullptr)       // (A/B and tests: every region through the serial chain)
                ICHK(f, hipMemsetAsync(f->d_wchained, 0, n_tasks * sizeof(WalkChained), f->wstream));
            walk_chain_kernel<<<(unsigned)n_tasks, LANES, 0, f->wstream>>>(v, d_tasks, d_chunks, d_rec_base, f->d_wrecs, f->d_wchained);
            ICHK(f, hipGetLastError());
            if (total_recs > 0) {
                walk_parse_kernel<<<(unsigned)((total_recs + 255) / 256), 256, 0, f->wstream>>>(v, (int)n_tasks, d_rec_base, f->d_wrecs, f->d_wchained, f->d_wfields);
                ICHK(f, hipGetLastError());
            }
            pair_walk_kernel<<<(unsigned)n_tasks, PW_THREADS, walk_lds_bytes(table_cap), f->wstream>>>(v, d_tasks, d_rec_base, f->d_wrecs, f->d_wfields,
                f->d_wchained, (tredgpu_walk_result*)f->d_wres, f->d_wpairs,
                f->d_gpool, w->cap_global, f->d_tpool, w->cap_target, (unsigned long long*)(f->d_wres + n_tasks * sizeof(tredgpu_walk_result)),
                table_cap);
            ICHK(f, hipGetLastError());
        }
        ICHK(f, hipEventRecord(f->w1, f->wstream));
        f->walk_timed = true;
        ICHK(f, hipMemcpyAsync(f->h_wres, f->d_wres, n_tasks * sizeof(tredgpu_walk_result) + 16, hipMemcpyDeviceToHost, f->wstream));
        // the alternative loci's walks on a stream of their own, beside the pair walks: 480 pair-walk wavefronts leave half
        // of the SIMDs without one, and one after the other the two launches were 4.1 + 1.9 ms of every call
        // (also wblk / the walk view they read: copied in on wstream -- astream waits for that copy below)
        if (n_alt > 0) {
            ICHK(f, hipStreamWaitEvent(f->astream, f->w0, 0));
            alt_walk_kernel<<<(unsigned)n_alt, LANES, WALK_WINDOW, f->astream>>>(v, (const tredgpu_walk_task*)f->d_atask,
                (const tredgpu_walk_chunk*)(f->d_atask + n_alt * sizeof(tredgpu_walk_task)), (tredgpu_alt_result*)f->d_ares,
                f->d_ares + n_alt * sizeof(tredgpu_alt_result));
            ICHK(f, hipGetLastError());
            ICHK(f, hipMemcpyAsync(f->h_ares, f->d_ares, n_alt * sizeof(tredgpu_alt_result) + nb, hipMemcpyDeviceToHost, f->astream));
        }
        ICHK(f, hipEventRecord(f->adone, f->astream));
        ICHK(f, hipEventRecord(f->wdone, f->wstream));
    }
    for (int s = 0; s < nstreams; ++s) {
        ICHK(f, hipEventRecord(f->t1[s], f->stream[s]));
        ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
    }
    tr.mark("enqueued");
    for (int s = 0; s < nstreams; ++s)
        if (wait_asleep(f, f->done[s])) return -10;
    tr.mark("decoded");
    int bad = 0;
    for (int32_t k = 0; k < n_blocks; ++k) { status[k] = f->h_status[k]; bad += status[k] != 0; }
    if (crc) for (int32_t k = 0; k < n_blocks; ++k) crc[k] = (uint32_t)f->h_status[f->cap_blocks + k];
    if (w) {
        if (wait_asleep(f, f->wdone)) return -10;
        tr.mark("pair_walk");
        if (wait_asleep(f, f->adone)) return -10;
        tr.mark("alt_walk");
        unsigned long long used[2];
        memcpy(used, f->h_wres + n_tasks * sizeof(tredgpu_walk_result), 16);
        memcpy(w->results, f->h_wres, n_tasks * sizeof(tredgpu_walk_result));
        if (n_alt > 0) {
            memcpy(w->alt_results, f->h_ares, n_alt * sizeof(tredgpu_alt_result));
            memcpy(w->need, f->h_ares + n_alt * sizeof(tredgpu_alt_result), (size_t)n_blocks);
        }
        // (a task that found its pool full took its room all the same: the counters can exceed the capacities)
        const size_t ng = (size_t)std::min<unsigned long long>(used[0], (unsigned long long)w->cap_global),
                     nt = (size_t)std::min<unsigned long long>(used[1], (unsigned long long)w->cap_target);
        for (int which = 0; which < 2; ++which) {
            int32_t** h = which ? &f->h_tpool : &f->h_gpool;
            size_t* cap = which ? &f->cap_htpool : &f->cap_hgpool;
            const size_t need = ((which ? nt : ng) + 16) * 4;
            if (need > *cap) {
                const size_t c = std::max(need, *cap + *cap / 8);
                if (*h) (void)hipHostFree(*h);
                *h = nullptr; *cap = 0;
                ICHK(f, hipHostMalloc((void**)h, c, hipHostMallocDefault));
                *cap = c;
            }
        }
        if (ng) ICHK(f, hipMemcpyAsync(f->h_gpool, f->d_gpool, ng * 4, hipMemcpyDeviceToHost, f->wstream));
        if (nt) ICHK(f, hipMemcpyAsync(f->h_tpool, f->d_tpool, nt * 4, hipMemcpyDeviceToHost, f->wstream));
        ICHK(f, hipEventRecord(f->wdone, f->wstream));
        if (wait_asleep(f, f->wdone)) return -10;
        if (ng) memcpy(w->global_pool, f->h_gpool, ng * 4);
        if (nt) memcpy(w->target_pool, f->h_tpool, nt * 4);
        w->n_global = (int64_t)ng;
        w->n_target = (int64_t)nt;
        tr.mark("pools");
    }
    return bad;
}
}  // namespace

extern "C" {

int tredgpu_inflate_blocks_crc(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc) {
    return run_inflate(f, n_blocks, status, crc, true, nullptr);
}

int tredgpu_inflate_walk(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc, tredgpu_walk_args* walk) {
    if (!walk) return f ? ifail(f, -2, "walk is NULL") : -2;
    return run_inflate(f, n_blocks, status, crc, false, walk);
}

// The blocks with need[k] != 0 of the last tredgpu_inflate_walk, copied to their places in the pinned output (runs of
// wanted blocks, and the unwanted ones between two runs when they are few, go in one copy).
int tredgpu_inflater_fetch(tredgpu_inflater* f, int32_t n_blocks, const uint8_t* need) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || (n_blocks > 0 && !need)) return ifail(f, -2, "bad arguments");
    if (n_blocks == 0) return 0;
    if (!f->h_out) return ifail(f, -2, "this inflater keeps no host copy of the output: tredgpu_inflater_fetch_dense");
    const int64_t* ooff = f->h_off + f->cap_blocks;
    ICHK(f, hipSetDevice(f->device));
    constexpr int64_t GAP = 128 * 1024;           // a copy costs the host ~5 us: less than these bytes cost the bus
    int copies = 0;
    int32_t k = 0;
    while (k < n_blocks) {
        if (!need[k]) { ++k; continue; }
        int32_t last = k;                          // the run [k, last]
        for (int32_t j = k + 1; j < n_blocks && ooff[j] - ooff[last + 1] <= GAP; ++j)
            if (need[j]) last = j;
        ICHK(f, hipMemcpyAsync(f->h_out + ooff[k], f->d_out + ooff[k], (size_t)(ooff[last + 1] - ooff[k]), hipMemcpyDeviceToHost, f->stream[copies & 1]));
        ++copies;
        k = last + 1;
    }
    for (int s = 0; s < 2; ++s) {
        ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
        if (wait_asleep(f, f->done[s])) return -10;
    }
    return copies;
}

int tredgpu_inflate_blocks(tredgpu_inflater* f, int32_t n_blocks, int32_t* status) { return tredgpu_inflate_blocks_crc(f, n_blocks, status, nullptr); }

int tredgpu_inflater_timing(tredgpu_inflater* f, double* total_ms, double* kernel_ms) {
    if (!f || !total_ms || !kernel_ms) return -2;
    *total_ms = *kernel_ms = 0.0;
    if (f->last_slices == 0) return 0;
    ICHK(f, hipSetDevice(f->device));
    float ms = 0.f;
    for (int s = 0; s < f->last_streams; ++s) {
        ICHK(f, hipEventElapsedTime(&ms, f->t0[0], f->t1[s]));
        *total_ms = std::max(*total_ms, (double)ms);
    }
    for (int k = 0; k < f->last_slices; ++k) {
        if (hipEventElapsedTime(&ms, f->k0[k], f->k1[k]) == hipSuccess) *kernel_ms += ms;
    }
    return 0;
}

int tredgpu_inflater_host_out(tredgpu_inflater* f, int enabled) {
    if (!f) return -2;
    if ((enabled != 0) != f->host_out) {
        ICHK(f, hipSetDevice(f->device));
        for (hipStream_t st : f->stream) ICHK(f, hipStreamSynchronize(st));
        release(f);                                    // (the next reserve allocates what the new mode needs)
        f->host_out = enabled != 0;
    }
    return 0;
}

int tredgpu_inflater_fetch_dense(tredgpu_inflater* f, int32_t n_blocks, const uint8_t* need, uint8_t** host, int64_t* dense_off) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || !host || !dense_off || (n_blocks > 0 && !need)) return ifail(f, -2, "bad arguments");
    CallTrace tr("fetch_dense");
    const int64_t* ooff = f->h_off + f->cap_blocks;
    constexpr int64_t GAP = 128 * 1024;           // a copy costs the host ~5 us: less than these bytes cost the bus
    // the runs: a wanted block, and on to the next wanted one while the blocks in between are fewer bytes than GAP
    struct Run { int32_t first, last; };
    std::vector<Run> runs;
    int64_t total = 0;
    dense_off[0] = 0;
    int32_t k = 0;
    while (k < n_blocks) {
        if (!need[k]) { dense_off[k + 1] = total; ++k; continue; }
        int32_t last = k;
        for (int32_t j = k + 1; j < n_blocks && ooff[j] - ooff[last + 1] <= GAP; ++j)
            if (need[j]) last = j;
        for (int32_t j = k; j <= last; ++j) { total += ooff[j + 1] - ooff[j]; dense_off[j + 1] = total; }
        runs.push_back(Run{k, last});
        k = last + 1;
    }
    ICHK(f, hipSetDevice(f->device));
    if ((size_t)total + 64 > f->cap_dense) {
        const size_t c = std::max((size_t)total + 64, f->cap_dense + f->cap_dense / 8);
        if (f->h_dense) (void)hipHostFree(f->h_dense);
        f->h_dense = nullptr; f->cap_dense = 0;
        ICHK(f, hipHostMalloc((void**)&f->h_dense, c, hipHostMallocDefault));
        f->cap_dense = c;
    }
    tr.mark("room");
    *host = f->h_dense;
    int copies = 0;
    static const bool by_dma = getenv("TREDGPU_FETCH_DMA") != nullptr;     // (A/B: the copy engines, one copy per run)
    if (by_dma) {
        for (const Run& r : runs) {
            ICHK(f, hipMemcpyAsync(f->h_dense + dense_off[r.first], f->d_out + ooff[r.first], (size_t)(ooff[r.last + 1] - ooff[r.first]),
                                   hipMemcpyDeviceToHost, f->stream[copies & 1]));
            ++copies;
        }
        tr.mark("enqueued");
        for (int s = 0; s < 2; ++s) {
            ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
            if (wait_asleep(f, f->done[s])) return -10;
        }
    } else if (!runs.empty()) {
        constexpr int64_t PIECE = 32 * 1024;       // bytes per workgroup: ~3 300 workgroups for a 16-sample call
        size_t n_pieces = 0;
        for (const Run& r : runs) n_pieces += (size_t)((ooff[r.last + 1] - ooff[r.first] + PIECE - 1) / PIECE);
        if (n_pieces > f->cap_pieces) {
            const size_t c = std::max(n_pieces, f->cap_pieces + f->cap_pieces / 2);
            if (f->h_pieces) (void)hipHostFree(f->h_pieces);
            if (f->d_pieces) (void)hipFree(f->d_pieces);
            f->h_pieces = f->d_pieces = nullptr; f->cap_pieces = 0;
            ICHK(f, hipHostMalloc((void**)&f->h_pieces, c * sizeof(FetchPiece), hipHostMallocDefault));
            ICHK(f, hipMalloc((void**)&f->d_pieces, c * sizeof(FetchPiece)));
            f->cap_pieces = c;
        }
        FetchPiece* P = (FetchPiece*)f->h_pieces;
        size_t p = 0;
        for (const Run& r : runs) {
            const int64_t bytes = ooff[r.last + 1] - ooff[r.first];
            for (int64_t o = 0; o < bytes; o += PIECE) P[p++] = FetchPiece{ooff[r.first] + o, dense_off[r.first] + o, (int32_t)std::min(PIECE, bytes - o), 0};
        }
        ICHK(f, hipMemcpyAsync(f->d_pieces, f->h_pieces, n_pieces * sizeof(FetchPiece), hipMemcpyHostToDevice, f->stream[0]));
        fetch_gather_kernel<<<(unsigned)n_pieces, 256, 0, f->stream[0]>>>(f->d_out, f->h_dense, (const FetchPiece*)f->d_pieces);
        ICHK(f, hipGetLastError());
        copies = (int)n_pieces;
        tr.mark("enqueued");
        ICHK(f, hipEventRecord(f->done[0], f->stream[0]));
        if (wait_asleep(f, f->done[0])) return -10;
    }
    tr.mark("copied");
    if (tr.on) { char b[64]; snprintf(b, sizeof b, " copies=%d MB=%.1f", copies, total / 1e6); tr.line += b; }
    return copies;
}

int64_t tredgpu_inflater_pinned_bytes(const tredgpu_inflater* f) {
    if (!f) return -2;
    size_t n = f->cap_comp + (f->h_out ? f->cap_out : 0) + f->cap_blocks * (2 * sizeof(int64_t) + 2 * sizeof(int32_t)) + f->cap_dense +
               f->cap_pieces * sizeof(FetchPiece) + f->cap_wblk + f->cap_wtask + f->cap_wres + f->cap_hgpool + f->cap_htpool + f->cap_atask + f->cap_ares;
    return (int64_t)n;
}

int64_t tredgpu_inflater_walk_serial_regions(tredgpu_inflater* f) {
    if (!f) return -2;
    if (f->last_walk_tasks == 0 || !f->d_wchained) return 0;
    ICHK(f, hipSetDevice(f->device));
    std::vector<WalkChained> c(f->last_walk_tasks);
    ICHK(f, hipMemcpy(c.data(), f->d_wchained, c.size() * sizeof(WalkChained), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (const WalkChained& w : c) n += w.mode == 0;
    if (getenv("TREDGPU_TRACE") != nullptr)
        for (size_t t = 0; t < c.size(); ++t)
            if (c[t].mode == 0) fprintf(stderr, "tredgpu: region %zu chained serially (reason %d), %d records, status %d\n", t, c[t].pad, c[t].n, c[t].status);
    return n;
}

int tredgpu_inflater_walk_ms(tredgpu_inflater* f, double* walk_ms) {
    if (!f || !walk_ms) return -2;
    *walk_ms = 0.0;
    if (!f->walk_timed) return 0;
    ICHK(f, hipSetDevice(f->device));
    float ms = 0.f;
    ICHK(f, hipEventElapsedTime(&ms, f->w0, f->w1));
    *walk_ms = ms;
    return 0;
}

}  // extern "C"
