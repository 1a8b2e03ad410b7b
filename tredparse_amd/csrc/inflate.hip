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
//     length code through a direct table in LDS (11 bits; longer codes canonically from the length counts), the
//     length's extra bits, the distance code (9-bit direct table) and its extra bits, all from the lane's own 57-bit
//     view of the stream (15 + 5 + 15 + 13 = 48 bits at most) -- and packs (bits consumed, kind, length or literal,
//     distance) into one dword.  63 of the 64 answers are for offsets no symbol starts at; they cost nothing but the
//     VALU slots of a wavefront that would otherwise idle behind a serial chain.
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
//   * Tables are built by the wavefront together: code-length histogram by LDS atomics, symbols ranked within their
//     length by ballot + prefix popcount, direct-table entries decoded canonically entry-parallel.
//   * LDS: 6.3 KB per wavefront (25 wavefronts per CU instead of one); no per-block workspace in global memory.
// Nothing here knows BAM: the C ABI (include/tredgpu.h, tredgpu_inflate_*) takes payload offsets and sizes and
// returns bytes plus a status per block; gzip framing and ISIZE stay with the host library (libtredbam).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <string>

#include "../../include/tredgpu.h"

namespace {

constexpr int LANES = 64;
constexpr int MAXBITS = 15, MAXL = 288, MAXD = 32;
#ifndef ROOTL_BITS
#define ROOTL_BITS 11
#define ROOTD_BITS 9
#endif
constexpr int ROOTL = ROOTL_BITS, ROOTD = ROOTD_BITS;

// one wavefront's tables (one block in flight per wavefront)
struct WaveLds {
    uint16_t fastL[1 << ROOTL];   // next ROOTL bits of the stream -> symbol | code length << 9 (0: a longer code, or none)
    uint16_t fastD[1 << ROOTD];
    uint16_t symL[MAXL];          // symbols sorted by (code length, symbol): canonical decoding of the longer codes
    uint16_t symD[MAXD];
    uint32_t cnt[16];             // codes per length of the alphabet under construction
    uint8_t lens[MAXL + MAXD];    // code lengths as the block header gives them
    uint8_t clsym[32];            // the code-length code: sorted symbols and its 7-bit direct table (symbol | length << 5)
    uint8_t clfast[128];
    uint32_t queue[2 * LANES];    // decoded symbols in stream order, waiting to be executed 64 at a time
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

// what canonical decoding needs for the codes longer than the direct table's ROOT bits (wave-uniform)
template <int ROOT>
struct LongCodes {
    int first, index;               // first code and symbol index at length ROOT + 1
    int count[MAXBITS - ROOT];      // codes of length ROOT + 1 .. 15
};

// count[] / sorted symbols / direct table of a canonical code from n code lengths in LDS, by the whole wavefront.
// Returns <0 for an over-subscribed set, >0 for an incomplete one, 0 for a complete one (puff's `left`); zeros = the
// number of unused symbols.
template <int ROOT>
__device__ int build_tables(WaveLds& S, const uint8_t* lens, int n, uint16_t* sym, uint16_t* fast, LongCodes<ROOT>& C, int& zeros, int lane) {
    if (lane < 16) S.cnt[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += LANES) atomicAdd(&S.cnt[lens[s]], 1u);
    __syncthreads();
    int c[MAXBITS + 1];
#pragma unroll
    for (int l = 0; l <= MAXBITS; ++l) c[l] = (int)S.cnt[l];
    zeros = c[0];
    int left = 1;
#pragma unroll
    for (int l = 1; l <= MAXBITS; ++l) {
        left <<= 1;
        left -= c[l];
        if (left < 0) return left;
    }
    // symbols sorted by (length, symbol): the rank inside a length is a ballot and a prefix popcount per 64 symbols
    int off[MAXBITS + 1];
    off[1] = 0;
#pragma unroll
    for (int l = 1; l < MAXBITS; ++l) off[l + 1] = off[l] + c[l];
    for (int s0 = 0; s0 < n; s0 += LANES) {
        const int s = s0 + lane;
        const int l = s < n ? (int)lens[s] : 0;
#pragma unroll
        for (int L = 1; L <= MAXBITS; ++L) {
            if (c[L] == 0) continue;
            const uint64_t m = __builtin_amdgcn_ballot_w64(l == L);
            if (l == L) sym[off[L] + lanes_below(m)] = (uint16_t)s;
            off[L] += (int)__popcll(m);
        }
    }
    __syncthreads();
    // the direct table, entry by entry: the entry's low bits in stream order are a code of at most ROOT bits (decoded
    // canonically, as the stream's bits would be) or the head of a longer one (0)
    for (int t = lane; t < (1 << ROOT); t += LANES) {
        int code = 0, first = 0, index = 0, found = -1, flen = 0;
#pragma unroll
        for (int len = 1; len <= ROOT; ++len) {
            code |= (t >> (len - 1)) & 1;
            const int count = c[len];
            if (found < 0 && code - count < first) { found = index + (code - first); flen = len; }
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        fast[t] = found < 0 ? (uint16_t)0 : (uint16_t)(sym[found] | flen << 9);
    }
    int first = 0, index = 0;
#pragma unroll
    for (int l = 1; l <= ROOT; ++l) { index += c[l]; first = (first + c[l]) << 1; }
    C.first = first;
    C.index = index;
#pragma unroll
    for (int l = ROOT + 1; l <= MAXBITS; ++l) C.count[l - ROOT - 1] = c[l];
    __syncthreads();
    return zeros == n ? 0 : left;        // no codes at all: complete, nothing decodes (as puff and zlib have it)
}

// the symbol whose code starts at bit 0 of `bits` (>= 15 valid bits): direct table, else canonically from length
// ROOT + 1 on; -1: no such code.  Per lane: every lane asks for another offset of the stream.
template <int ROOT>
__device__ __forceinline__ int decode_at(uint32_t bits, const uint16_t* fast, const uint16_t* sym, const LongCodes<ROOT>& C, int& clen) {
    const uint32_t e = fast[bits & ((1u << ROOT) - 1u)];
    if (e != 0) { clen = (int)(e >> 9); return (int)(e & 511u); }
    int code = (int)(__builtin_bitreverse32(bits) >> (32 - ROOT)) << 1;   // the first ROOT bits as a code, room for the next
    int first = C.first, index = C.index;
    uint32_t rest = bits >> ROOT;
    int found = -1, flen = 0;
#pragma unroll
    for (int len = ROOT + 1; len <= MAXBITS; ++len) {
        code |= (int)(rest & 1u);
        rest >>= 1;
        const int count = C.count[len - ROOT - 1];
        if (found < 0 && code - count < first) { found = index + (code - first); flen = len; }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    clen = flen;
    return found < 0 ? -1 : (int)sym[found];
}

// the complete symbol that starts at this lane's bit of the stream, packed
__device__ __forceinline__ uint32_t symbol_at(uint64_t view, const WaveLds& S, const LongCodes<ROOTL>& CL, const LongCodes<ROOTD>& CD) {
    int clen;
    const int sym = decode_at<ROOTL>((uint32_t)view, S.fastL, S.symL, CL, clen);
    if (sym < 0) return BAD_SYMBOL;
    if (sym < 256) return (uint32_t)clen << P_BITS | K_LIT << P_KIND | (uint32_t)sym;
    if (sym == 256) return 64u << P_BITS | K_END << P_KIND | (uint32_t)clen;
    const int c = sym - 257;
    if (c >= 29) return BAD_SYMBOL;
    int base, extra;
    len_code(c, base, extra);
    const int mlen = base + (int)((uint32_t)(view >> clen) & ((1u << extra) - 1u));
    const int used = clen + extra;                        // <= 20
    const uint64_t v2 = view >> used;                     // >= 37 valid bits left
    int dlen;
    const int ds = decode_at<ROOTD>((uint32_t)v2, S.fastD, S.symD, CD, dlen);
    if (ds < 0 || ds >= 30) return BAD_SYMBOL;
    dist_code(ds, base, extra);
    const int dist = base + (int)((uint32_t)(v2 >> dlen) & ((1u << extra) - 1u));
    return (uint32_t)(used + dlen + extra) << P_BITS | K_MATCH << P_KIND | (uint32_t)(dist - 1) << P_DIST | (uint32_t)(mlen - 3);
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
    // a match whose source ends before the batch begins depends on nothing in the batch
    const bool own = is_match && dst - dist + len <= opos && len <= 16;
    if (own) {
        const uint8_t* src = o + dst - dist;
        uint8_t* d = o + dst;
        const uint64_t lo = *reinterpret_cast<const U64*>(src);
        if (len >= 8) {
            const uint64_t hi = *reinterpret_cast<const U64*>(src + 8);
            const int sh = (len - 8) * 8;                  // bytes [len - 8, len) of hi:lo
            const uint64_t tail = sh == 0 ? lo : (sh == 64 ? hi : (lo >> sh) | (hi << (64 - sh)));
            *reinterpret_cast<U64*>(d) = lo;
            *reinterpret_cast<U64*>(d + len - 8) = tail;
        } else if (len >= 4) {
            *reinterpret_cast<U32*>(d) = (uint32_t)lo;
            *reinterpret_cast<U32*>(d + len - 4) = (uint32_t)(lo >> ((len - 4) * 8));
        } else {                                           // 3
            *reinterpret_cast<U16*>(d) = (uint16_t)lo;
            d[2] = (uint8_t)(lo >> 16);
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

__global__ __launch_bounds__(LANES) void inflate_kernel(const uint32_t* __restrict__ comp, const int64_t* __restrict__ comp_off,
                                                        uint8_t* out, const int64_t* __restrict__ out_off, int n_blocks,
                                                        int32_t* __restrict__ status) {
    __shared__ WaveLds S;
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    const int64_t c0 = comp_off[g], c1 = comp_off[g + 1];
    uint8_t* o = out + out_off[g];
    const int olen = (int)(out_off[g + 1] - out_off[g]);
    const uint32_t* p = comp + (c0 >> 2);
    const uint8_t* p8 = reinterpret_cast<const uint8_t*>(p);
    const int nbytes = (int)(c1 - c0);
    const int nwords = (nbytes + 3) >> 2;
    int bit = 0, opos = 0, rc = nbytes > 0 ? 0 : -1, last = 0;
    LongCodes<ROOTL> CL = {};
    LongCodes<ROOTD> CD = {};
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
            for (int s = lane; s < MAXL; s += LANES) S.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            if (lane < MAXD) S.lens[MAXL + lane] = 5;
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
            if (mycl != 0) S.clsym[myrank] = (uint8_t)lane;
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
                S.clfast[t] = found < 0 ? (uint8_t)0 : (uint8_t)(S.clsym[found] | flen << 5);
            }
            __syncthreads();
            // the nlen + ndist code lengths, run-length coded
            int idx = 0, prev = 0;
            const int total = nlen + ndist;
            while (idx < total) {
                const uint32_t e = S.clfast[b.peek(7)];
                if (e == 0) { rc = -1; break; }
                b.skip((int)(e >> 5));
                const int sym = (int)(e & 31u);
                if (sym < 16) {
                    if (lane == 0) S.lens[idx] = (uint8_t)sym;
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
                    for (int j = lane; j < rep; j += LANES) S.lens[idx + j] = (uint8_t)prev;
                    idx += rep;
                }
            }
            if (rc != 0) break;
            __syncthreads();
            if (S.lens[256] == 0) { rc = -1; break; }      // no end-of-block code
            // the distance lengths follow the literal/length lengths directly: move them to their own place
            const int dl = lane < ndist ? (int)S.lens[nlen + lane] : 0;
            __syncthreads();
            if (lane < MAXD) S.lens[MAXL + lane] = (uint8_t)dl;
        }
        bit = b.pos();
        __syncthreads();
        int zeros;
        int err = build_tables<ROOTL>(S, S.lens, nlen, S.symL, S.fastL, CL, zeros, lane);
        if (err < 0 || (err > 0 && nlen - zeros != 1)) { rc = -1; break; }
        err = build_tables<ROOTD>(S, S.lens + MAXL, ndist, S.symD, S.fastD, CD, zeros, lane);
        if (err < 0 || (err > 0 && ndist - zeros != 1)) { rc = -1; break; }

        // ---- the block's symbols ----
        // window = the 64 bit offsets bit0 .. bit0 + 63; pos = where the next symbol starts, relative to bit0
        int bit0 = bit, pos = 0, nsym = 0, end = 0;
        uint64_t raw = *reinterpret_cast<const U64*>(p8 + ((bit0 + lane) >> 3));
        while (!end) {
            if (bit0 > nbytes * 8) { rc = -1; break; }     // a symbol would start behind the payload
            const uint64_t view = raw >> ((bit0 + lane) & 7);
            raw = *reinterpret_cast<const U64*>(p8 + ((bit0 + LANES + lane) >> 3));   // the next window's bytes, early
            const uint32_t sp = symbol_at(view, S, CL, CD);
            // the walk: from symbol start to symbol start, on the scalar unit
            uint64_t starts = 0;
            uint32_t e;
            int at;
            do {
                at = pos;
                e = (uint32_t)__builtin_amdgcn_readlane((int)sp, at);
                starts |= 1ull << at;
                pos += (int)(e >> P_BITS);
            } while (pos < LANES);
            const uint32_t kind = (e >> P_KIND) & 3u;
            if (kind >= K_END) {                           // the last symbol of the block (or nothing decodable): not queued
                if (kind == K_BAD) { rc = -1; break; }
                starts &= ~(1ull << at);
                bit = bit0 + at + (int)(e & 255u);
                end = 1;
            }
            // the lanes that hold a real symbol append it to the queue, in stream order
            if ((starts >> lane) & 1ull) S.queue[nsym + lanes_below(starts)] = sp;
            nsym += (int)__popcll(starts);
            __syncthreads();
            while (nsym >= LANES || (end && nsym > 0)) {
                const int n = min(nsym, LANES);
                const uint32_t q = S.queue[lane];
                const uint32_t q2 = S.queue[LANES + lane];
                if (run_queue(o, olen, opos, q, n, lane) != 0) { rc = -1; end = 1; break; }
                nsym -= n;
                __syncthreads();
                if (lane < nsym) S.queue[lane] = q2;
                __syncthreads();
            }
            pos -= LANES;
            bit0 += LANES;
        }
        if (rc != 0) break;
        __syncthreads();                                   // the tables are rebuilt by the next header
    }
    if (rc == 0) {
        if (opos != olen) rc = -2;                         // fewer bytes than the trailer's ISIZE
        else if (bit > nbytes * 8) rc = -1;                // ran past the payload
    }
    if (lane == 0) status[g] = rc;
}

}  // namespace

// ---- C ABI (include/tredgpu.h) ------------------------------------------------------------------------------------
struct tredgpu_inflater {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;          // created with hipEventBlockingSync: waiting for a call sleeps, it does not spin
    uint8_t *h_comp = nullptr, *h_out = nullptr;      // pinned staging the caller fills / reads in place
    int64_t *h_off = nullptr;                         // pinned: comp_off[n+1] then out_off[n+1]
    int32_t* h_status = nullptr;
    size_t cap_comp = 0, cap_out = 0, cap_blocks = 0;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    int64_t* d_off = nullptr;
    int32_t* d_status = nullptr;
    std::string err;
};

namespace {
thread_local std::string g_inflate_error;

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
    if (f->h_off) (void)hipHostFree(f->h_off);
    if (f->h_status) (void)hipHostFree(f->h_status);
    for (void* p : {(void*)f->d_comp, (void*)f->d_out, (void*)f->d_off, (void*)f->d_status})
        if (p) (void)hipFree(p);
    f->h_comp = f->h_out = nullptr; f->h_off = nullptr; f->h_status = nullptr;
    f->d_comp = f->d_out = nullptr; f->d_off = nullptr; f->d_status = nullptr;
    f->cap_comp = f->cap_out = f->cap_blocks = 0;
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
    // several of these (kernels of different streams run one after the other here, and one of these takes 20-40 ms)
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, lo_prio)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&f->done, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) {
        if (f->stream) (void)hipStreamDestroy(f->stream);
        delete f;
        return ifail(nullptr, -10, "stream / event creation", e);
    }
    *out = f;
    return 0;
}

void tredgpu_inflater_destroy(tredgpu_inflater* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(f->stream);
    release(f);
    (void)hipEventDestroy(f->done);
    (void)hipStreamDestroy(f->stream);
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
        ICHK(f, hipStreamSynchronize(f->stream));
        const size_t cc = std::max(need_c, f->cap_comp + f->cap_comp / 2), co = std::max(need_o, f->cap_out + f->cap_out / 2),
                     cb = std::max(need_b, f->cap_blocks + f->cap_blocks / 2);
        release(f);
        ICHK(f, hipHostMalloc((void**)&f->h_comp, cc, hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_out, co, hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_off, 2 * cb * sizeof(int64_t), hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_status, cb * sizeof(int32_t), hipHostMallocDefault));
        ICHK(f, hipMalloc((void**)&f->d_comp, cc));
        ICHK(f, hipMalloc((void**)&f->d_out, co));
        ICHK(f, hipMalloc((void**)&f->d_off, 2 * cb * sizeof(int64_t)));
        ICHK(f, hipMalloc((void**)&f->d_status, cb * sizeof(int32_t)));
        f->cap_comp = cc; f->cap_out = co; f->cap_blocks = cb;
    }
    *comp_host = f->h_comp;
    *out_host = f->h_out;
    *comp_off_host = f->h_off;
    *out_off_host = f->h_off + f->cap_blocks;
    return 0;
}

int tredgpu_inflate_blocks(tredgpu_inflater* f, int32_t n_blocks, int32_t* status) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || (n_blocks > 0 && !status)) return ifail(f, -2, "bad arguments (reserve first)");
    if (n_blocks == 0) return 0;
    const int64_t* coff = f->h_off;
    const int64_t* ooff = f->h_off + f->cap_blocks;
    for (int32_t k = 0; k < n_blocks; ++k) {
        if (coff[k] < 0 || (coff[k] & 3) != 0 || coff[k + 1] < coff[k] || ooff[k] < 0 || ooff[k + 1] < ooff[k] ||
            ooff[k + 1] - ooff[k] > 65536)
            return ifail(f, -2, "block offsets: payloads start on 4-byte boundaries, ascend, and inflate to at most 64 KiB each");
    }
    if ((size_t)coff[n_blocks] + 64 > f->cap_comp || (size_t)ooff[n_blocks] + 64 > f->cap_out) return ifail(f, -2, "offsets beyond the reserved buffers");
    ICHK(f, hipSetDevice(f->device));
    const size_t cbytes = ((size_t)coff[n_blocks] + 3) & ~(size_t)3;
    ICHK(f, hipMemcpyAsync(f->d_comp, f->h_comp, cbytes, hipMemcpyHostToDevice, f->stream));
    ICHK(f, hipMemcpyAsync(f->d_off, f->h_off, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
    ICHK(f, hipMemcpyAsync(f->d_off + f->cap_blocks, f->h_off + f->cap_blocks, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream));
    inflate_kernel<<<n_blocks, LANES, 0, f->stream>>>((const uint32_t*)f->d_comp, f->d_off, f->d_out, f->d_off + f->cap_blocks, n_blocks,
                                                     f->d_status);
    ICHK(f, hipGetLastError());
    ICHK(f, hipMemcpyAsync(f->h_out, f->d_out, (size_t)ooff[n_blocks], hipMemcpyDeviceToHost, f->stream));
    ICHK(f, hipMemcpyAsync(f->h_status, f->d_status, (size_t)n_blocks * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
    ICHK(f, hipEventRecord(f->done, f->stream));
    // wait asleep: hipEventSynchronize spins even on a hipEventBlockingSync event here (measured: CPU time = wall time,
    // and calls of other threads on other streams queue up behind the spinning one); the host threads that wait are the
    // ones whose cores the path is short of, and a call takes tens of milliseconds
    for (;;) {
        const hipError_t q = hipEventQuery(f->done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return ifail(f, -10, "hipEventQuery", q);
        usleep(200);
    }
    int bad = 0;
    for (int32_t k = 0; k < n_blocks; ++k) { status[k] = f->h_status[k]; bad += status[k] != 0; }
    return bad;
}

}  // extern "C"
