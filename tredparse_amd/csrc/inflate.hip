// inflate.hip -- raw DEFLATE (RFC 1951) decoding of many BGZF blocks at once on gfx950.
//
// The read-selection front end of the path (/root/reference/tredparse/bam_parser.py:184-257, 316-369: pysam fetch /
// pileup, i.e. htslib's bgzf_read -> zlib inflate) spends two thirds of its host time inflating BGZF blocks: 35 MB
// per 30x sample, 31 of the 42 ms of tredbam_scan, on a box whose 16 host cores -- not its GPU, which idles -- bound the
// end-to-end rate.  A BGZF block is at most 64 KiB and independent of every other block, and a sample needs ~550 of
// them: here ONE LANE decodes ONE BLOCK, 64 blocks per wavefront, thousands of blocks in flight per call.
//
//   * per lane a little state machine -- block header / symbol / match copy / done -- advanced one step per loop
//     trip, so that lanes in different states cost each other one short step each, not a whole match copy or a whole
//     header;
//   * canonical Huffman decoding straight from the code-length counts (count[len], symbols sorted by code), bit by
//     bit (704 bytes of tables per lane), behind a direct table on the next 9 bits of the stream (7 for distances):
//     2 KB per lane, held in LDS lane-minor ([entry][lane]) -- 127 KB per wavefront, one wavefront per CU; the two-level
//     tables of the host decoder (csrc/inflate_block.h) would not fit per lane;
//   * the compressed payloads start on 4-byte boundaries of the staging buffer (the host lays them out) and are
//     read as aligned dwords into a 64-bit bit buffer.
// Nothing here knows BAM: the C ABI (include/tredgpu.h, tredgpu_inflate_*) takes payload offsets and sizes and
// returns bytes plus a status per block; gzip framing, CRC-32 and ISIZE stay with the host library (libtredbam).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <string>

#include "../../include/tredgpu.h"

namespace {

constexpr int LANES = 64;
constexpr int MAXBITS = 15, MAXL = 288, MAXD = 32, MAXLENS = 320;
// uint16 entries per lane: count[16] + symbol[288] for literal/length codes, count[16] + symbol[32] for distances,
// then per code where canonical decoding resumes behind the direct table (first code and symbol index at length
// FAST + 1) and the direct tables: the next FAST bits of the stream -> symbol | length << 9 (0: a longer code)
// (7 / 4 bits make it 64 KB per wavefront and two wavefronts per CU: 809 samples/s with 56 samples per launch against
//  664, but 608 against 660 with 28, and 25 ms instead of 21 for a single sample)
#ifndef FASTL_BITS
#define FASTL_BITS 9
#define FASTD_BITS 7
#endif
constexpr int FASTL = FASTL_BITS, FASTD = FASTD_BITS;
constexpr int T_LCNT = 0, T_LSYM = 16, T_DCNT = 16 + MAXL, T_DSYM = 32 + MAXL, T_LCONT = 32 + MAXL + MAXD, T_DCONT = T_LCONT + 2,
              T_LFAST = T_DCONT + 2, T_DFAST = T_LFAST + (1 << FASTL), T_ENTRIES = T_DFAST + (1 << FASTD);
static_assert(T_ENTRIES * LANES * 2 <= 160 * 1024, "one wavefront's tables fit the CU's LDS");

// base value and extra bits of length code c (0..28) / distance code d (0..29), RFC 1951 3.2.5, in closed form (a
// table look-up per match would be a memory latency in every lane's way)
__device__ __forceinline__ void len_code(int c, int& base, int& extra) {
    extra = c < 8 || c == 28 ? 0 : (c >> 2) - 1;
    base = c < 8 ? 3 + c : (c == 28 ? 258 : 3 + ((4 + (c & 3)) << extra));
}
__device__ __forceinline__ void dist_code(int d, int& base, int& extra) {
    extra = d < 4 ? 0 : (d >> 1) - 1;
    base = d < 4 ? 1 + d : 1 + ((2 + (d & 1)) << extra);
}
__constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Bits {
    const uint32_t* p;
    int idx, nwords;      // next dword to fetch, dwords that hold payload
    uint64_t buf;
    int cnt;
    uint32_t ahead;       // dword idx - 1, fetched one refill early: its latency passes while the bits before it are used
    __device__ __forceinline__ void start(const uint32_t* at, int words) {
        p = at; nwords = words; buf = 0; cnt = 0;
        ahead = at[0];
        idx = 1;
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            // past the payload: zeros (the overrun is caught at the end).  The fetch itself is unconditional -- the staging
            // buffer has 64 bytes of slack -- so that the loaded dword lands in `ahead`'s own register: behind a select the
            // compiler loaded into a temporary, and the move out of it waited for the load on the spot
            const uint32_t w = idx - 1 < nwords ? ahead : 0u;
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            ahead = p[min(idx, nwords)];       // (clamped: a damaged stream may ask for bits far behind its payload)
            ++idx;
        }
    }
    __device__ __forceinline__ uint32_t get(int n) {   // n <= 16
        refill();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
    __device__ __forceinline__ long long consumed_bits() const { return (long long)(idx - 1) * 32 - cnt; }
};

// one symbol of a canonical code: the direct table on the next FB bits, else canonical decoding from length FB + 1
// on (count[len] at tab[(cnt0 + len) * LANES], symbols sorted by code at tab[(sym0 + k) * LANES]); -1: no such code
// What canonical decoding needs for the codes longer than the direct table's FB bits, in registers (the kernel runs one
// wavefront per SIMD: registers are free, and with 64 lanes some lane has a long code in most trips; from LDS the six
// counts of a literal/length look-up were a chain of six LDS latencies -- worth 2 % of the kernel, 21.0 -> 20.6 ms)
template <int FB>
struct LongCodes {
    int first, index;            // first code and symbol index at length FB + 1
    int count[MAXBITS - FB];     // codes of length FB + 1 .. 15
};

template <int FB>
__device__ __forceinline__ int decode(Bits& b, const uint16_t* tab, int fast0, const LongCodes<FB>& C, int sym0) {
    b.refill();
    const uint32_t bits = (uint32_t)b.buf;
    const uint32_t e = tab[(fast0 + (int)(bits & ((1u << FB) - 1u))) * LANES];
    if (e != 0) {
        const int len = (int)(e >> 9);
        b.buf >>= len;
        b.cnt -= len;
        return (int)(e & 511u);
    }
    int code = (int)(__builtin_bitreverse32(bits) >> (32 - FB)) << 1;   // the first FB bits as a code, room for the next
    int first = C.first, index = C.index;
    uint32_t rest = bits >> FB;
    int found = -1, flen = 0;
#pragma unroll
    for (int len = FB + 1; len <= MAXBITS; ++len) {
        code |= (int)(rest & 1u);
        rest >>= 1;
        const int count = C.count[len - FB - 1];
        if (found < 0 && code - count < first) { found = index + (code - first); flen = len; }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    if (found < 0) return -1;
    b.buf >>= flen;
    b.cnt -= flen;
    return tab[(sym0 + found) * LANES];
}

// the code-length code (19 symbols, at most 7 bits): plain canonical decoding, bit by bit
__device__ __forceinline__ int decode_slow(Bits& b, const uint16_t* tab, int cnt0, int sym0) {
    b.refill();
    int code = 0, first = 0, index = 0;
    uint32_t bits = (uint32_t)b.buf;
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = tab[(cnt0 + len) * LANES];
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            return tab[(sym0 + index + (code - first)) * LANES];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// count[] and symbol[] of a canonical code from n code lengths (puff's construct); returns <0 for an over-subscribed
// set, >0 for an incomplete one, 0 for a complete one
__device__ int construct(uint16_t* tab, int cnt0, int sym0, const uint8_t* lens, int n) {
    for (int l = 0; l <= MAXBITS; ++l) tab[(cnt0 + l) * LANES] = 0;
    for (int s = 0; s < n; ++s) ++tab[(cnt0 + lens[s]) * LANES];
    int left = 1;
    for (int l = 1; l <= MAXBITS; ++l) {
        left <<= 1;
        left -= tab[(cnt0 + l) * LANES];
        if (left < 0) return left;
    }
    uint16_t offs[MAXBITS + 1];
    offs[1] = 0;
    for (int l = 1; l < MAXBITS; ++l) offs[l + 1] = (uint16_t)(offs[l] + tab[(cnt0 + l) * LANES]);
    for (int s = 0; s < n; ++s)
        if (lens[s] != 0) tab[(sym0 + offs[lens[s]]++) * LANES] = (uint16_t)s;
    return left;
}

// the direct table of a code already constructed: every code of at most FB bits fills the 2^(FB - len) entries whose
// low bits are its bits in stream order; and where canonical decoding resumes for the longer ones
template <int FB>
__device__ void construct_fast(uint16_t* tab, int fast0, LongCodes<FB>& C, int cnt0, const uint8_t* lens, int n) {
    for (int k = 0; k < (1 << FB); ++k) tab[(fast0 + k) * LANES] = 0;
    uint16_t next[FB + 2];
    int first = 0, index = 0;
    for (int l = 1; l <= FB; ++l) {
        next[l] = (uint16_t)first;
        const int count = tab[(cnt0 + l) * LANES];
        index += count;
        first = (first + count) << 1;
    }
    C.first = first;
    C.index = index;
#pragma unroll
    for (int l = FB + 1; l <= MAXBITS; ++l) C.count[l - FB - 1] = tab[(cnt0 + l) * LANES];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l == 0 || l > FB) continue;
        const uint32_t code = next[l]++;
        const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
        const uint16_t e = (uint16_t)(s | l << 9);
        for (uint32_t k = r; k < (1u << FB); k += 1u << l) tab[(fast0 + (int)k) * LANES] = e;
    }
}

// 8 bytes at any address (global memory takes unaligned accesses on gfx950)
typedef uint64_t __attribute__((aligned(1))) U64;

enum : int { ST_HDR = 0, ST_SYM = 1, ST_COPY = 2, ST_DONE = 3, ST_ERR = 4 };

__global__ __launch_bounds__(LANES) void inflate_kernel(const uint32_t* __restrict__ comp, const int64_t* __restrict__ comp_off,
                                                        uint8_t* out, const int64_t* __restrict__ out_off, int n_blocks,
                                                        uint8_t* lens_ws, int32_t* __restrict__ status) {
    __shared__ uint16_t tables[T_ENTRIES * LANES];
    const int lane = threadIdx.x;
    const int g = blockIdx.x * LANES + lane;
    if (g >= n_blocks) return;
    uint16_t* tab = tables + lane;
    uint8_t* lens = lens_ws + (size_t)g * MAXLENS;
    const int64_t c0 = comp_off[g], c1 = comp_off[g + 1];
    uint8_t* o = out + out_off[g];
    const int olen = (int)(out_off[g + 1] - out_off[g]);
    Bits b;
    b.start(comp + (c0 >> 2), (int)((c1 - c0 + 3) >> 2));
    int st = (c1 - c0) > 0 ? ST_HDR : ST_ERR;
    int opos = 0, last = 0, mlen = 0, mdist = 0, mspan = 0;
    LongCodes<FASTL> longL = {};
    LongCodes<FASTD> longD = {};
    while (st < ST_DONE) {
        // (the three states are tried one after the other in every trip: a match decoded in this trip makes its first
        //  copy step in it -- most matches of a BAM block are shorter than one step --, an end-of-block code goes on to
        //  the next header: 25.5 -> 21.4 ms per 556-block sample.  Several literals per trip were slower: 24 ms with two,
        //  27 with three -- the lanes that have a match wait)
        if (st == ST_SYM) {
            int sym = decode<FASTL>(b, tab, T_LFAST, longL, T_LSYM);
            if (sym < 0) st = ST_ERR;
            else if (sym < 256) {
                if (opos < olen) o[opos++] = (uint8_t)sym; else st = ST_ERR;
            } else if (sym == 256) st = last ? ST_DONE : ST_HDR;
            else {
                sym -= 257;
                if (sym >= 29) st = ST_ERR;
                else {
                    int base, extra;
                    len_code(sym, base, extra);
                    mlen = base + (int)b.get(extra);
                    const int ds = decode<FASTD>(b, tab, T_DFAST, longD, T_DSYM);
                    if (ds < 0 || ds >= 30) st = ST_ERR;
                    else {
                        dist_code(ds, base, extra);
                        mdist = base + (int)b.get(extra);   // (extra bits <= 13)
                        mspan = mdist;
                        st = (mdist > opos || opos + mlen > olen) ? ST_ERR : ST_COPY;
                    }
                }
            }
        }
        if (st == ST_COPY) {
            // up to 16 bytes of the match per trip, taken from `mspan` bytes back: mspan is a multiple of the distance
            // (the bytes since opos - distance repeat with that period), starts as the distance itself and doubles
            // with every full span copied until it covers a trip -- source and destination of one trip never overlap,
            // so the trip is two independent wide loads and two wide stores instead of a chain of byte loads each
            // waiting for the store before it (which was 90 % of the kernel's time: some lane of the 64 is in a match
            // in nearly every trip).  (Writing the 16 bytes a trip later, so that the loads' latency passes under the next
            // trip's decoding, was slower: 23.5 ms -- the exact-length stores and the shared vmcnt cost more.)
            const int n = min(min(mlen, 16), mspan);
            uint8_t* dst = o + opos;
            const uint8_t* src = dst - mspan;
            if (opos + 16 <= olen) {               // (room to write 16 bytes whatever n is: the rest is overwritten later)
                const U64 a = *reinterpret_cast<const U64*>(src), c = *reinterpret_cast<const U64*>(src + 8);
                *reinterpret_cast<U64*>(dst) = a;
                *reinterpret_cast<U64*>(dst + 8) = c;
            } else {
                for (int k = 0; k < n; ++k) dst[k] = src[k];
            }
            if (n == mspan && mspan < 16) mspan += n;
            opos += n;
            mlen -= n;
            if (mlen == 0) st = ST_SYM;
        } else if (st == ST_HDR) {   // a deflate block header (and, for a stored block, its bytes)
            last = (int)b.get(1);
            const int type = (int)b.get(2);
            if (type == 0) {
                const int drop = b.cnt & 7;            // to the next byte boundary of the stream
                b.buf >>= drop;
                b.cnt -= drop;
                const uint32_t len = b.get(16), nlen = b.get(16);
                if ((len ^ 0xffffu) != nlen || opos + (int)len > olen) st = ST_ERR;
                else {
                    for (uint32_t k = 0; k < len; ++k) o[opos++] = (uint8_t)b.get(8);
                    st = last ? ST_DONE : ST_HDR;
                }
            } else if (type == 1) {
                for (int s = 0; s < 144; ++s) lens[s] = 8;
                for (int s = 144; s < 256; ++s) lens[s] = 9;
                for (int s = 256; s < 280; ++s) lens[s] = 7;
                for (int s = 280; s < MAXL; ++s) lens[s] = 8;
                construct(tab, T_LCNT, T_LSYM, lens, MAXL);
                construct_fast<FASTL>(tab, T_LFAST, longL, T_LCNT, lens, MAXL);
                for (int s = 0; s < 30; ++s) lens[s] = 5;
                construct(tab, T_DCNT, T_DSYM, lens, 30);
                construct_fast<FASTD>(tab, T_DFAST, longD, T_DCNT, lens, 30);
                st = ST_SYM;
            } else if (type == 2) {
                const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
                if (nlen > 286 || ndist > 30) st = ST_ERR;
                else {
                    for (int k = 0; k < 19; ++k) lens[k] = 0;
                    for (int k = 0; k < ncode; ++k) lens[CL_ORDER[k]] = (uint8_t)b.get(3);
                    // the code-length code uses the literal/length table's room for a moment
                    int err = construct(tab, T_LCNT, T_LSYM, lens, 19);
                    if (err != 0) st = ST_ERR;   // complete code required (as zlib does)
                    int idx = 0;
                    while (st != ST_ERR && idx < nlen + ndist) {
                        int sym = decode_slow(b, tab, T_LCNT, T_LSYM);
                        if (sym < 0) { st = ST_ERR; break; }
                        if (sym < 16) lens[idx++] = (uint8_t)sym;
                        else {
                            int prev = 0, rep;
                            if (sym == 16) {
                                if (idx == 0) { st = ST_ERR; break; }
                                prev = lens[idx - 1];
                                rep = 3 + (int)b.get(2);
                            } else if (sym == 17) rep = 3 + (int)b.get(3);
                            else rep = 11 + (int)b.get(7);
                            if (idx + rep > nlen + ndist) { st = ST_ERR; break; }
                            while (rep--) lens[idx++] = (uint8_t)prev;
                        }
                    }
                    if (st != ST_ERR) {
                        if (lens[256] == 0) st = ST_ERR;   // no end-of-block code
                        else {
                            // the distance lengths first: construct() of the literal/length code overwrites nothing of them
                            // (lens is in global memory), but the tables of the code-length code are dead from here on
                            err = construct(tab, T_LCNT, T_LSYM, lens, nlen);
                            if (err < 0 || (err > 0 && nlen - tab[(T_LCNT + 0) * LANES] != 1)) st = ST_ERR;
                            else {
                                err = construct(tab, T_DCNT, T_DSYM, lens + nlen, ndist);
                                if (err < 0 || (err > 0 && ndist - tab[(T_DCNT + 0) * LANES] != 1)) st = ST_ERR;
                                else {
                                    construct_fast<FASTL>(tab, T_LFAST, longL, T_LCNT, lens, nlen);
                                    construct_fast<FASTD>(tab, T_DFAST, longD, T_DCNT, lens + nlen, ndist);
                                    st = ST_SYM;
                                }
                            }
                        }
                    }
                }
            } else st = ST_ERR;
        }
    }
    int rc = 0;
    if (st == ST_ERR) rc = -1;
    else if (opos != olen) rc = -2;                                       // fewer bytes than the trailer's ISIZE
    else if (b.consumed_bits() > (long long)(c1 - c0) * 8) rc = -1;       // ran past the payload
    status[g] = rc;
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
    uint8_t *d_comp = nullptr, *d_out = nullptr, *d_lens = nullptr;
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
    for (void* p : {(void*)f->d_comp, (void*)f->d_out, (void*)f->d_lens, (void*)f->d_off, (void*)f->d_status})
        if (p) (void)hipFree(p);
    f->h_comp = f->h_out = nullptr; f->h_off = nullptr; f->h_status = nullptr;
    f->d_comp = f->d_out = f->d_lens = nullptr; f->d_off = nullptr; f->d_status = nullptr;
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
        ICHK(f, hipMalloc((void**)&f->d_lens, cb * MAXLENS));
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
    inflate_kernel<<<(n_blocks + LANES - 1) / LANES, LANES, 0, f->stream>>>((const uint32_t*)f->d_comp, f->d_off, f->d_out, f->d_off + f->cap_blocks,
                                                                          n_blocks, f->d_lens, f->d_status);
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
