// walk.hip -- the record walks over the blocks inflate_decode.hip wrote (PEextractor's pair lengths, the alternative loci's mate
// rescue) and the gather kernel that hands the host the blocks it still reads.  tredgpu_inflate_walk (inflater_api.hip) launches them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inflater_internal.h"

using namespace tredgpu_front;

namespace {

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
constexpr int RESOLVE_MAX_RECORDS = 32768;         // (the per-batch arrays below; 2 x WALK_PAIR_CAP names' worth of records)
constexpr int RESOLVE_MAX_BATCHES = RESOLVE_MAX_RECORDS / LANES, PAIR_MAX_BATCHES = WALK_PAIR_CAP / LANES;
constexpr uint32_t REC_NONE = 0x7FFFFu;
constexpr int REC_BITS = 19;

__device__ inline bool walk_todo(const tredgpu_walk_task& T, const WalkFields& F, bool mine) {
    const bool off_region = F.rtid != T.tid || F.rpos >= T.end;             // (rtid < T.tid: the chain ends at the others)
    const int64_t e = (F.rend < 0 || F.rend <= F.rpos) ? (int64_t)F.rpos + 1 : (int64_t)F.rend;
    // (T.span <= 0: a plain region whose records are only counted -- tredgpu_select_task with n_alt < 0 --, no pairs)
    return mine && T.span > 0 && !off_region && e > T.start && (F.flag & 0x1) && !(F.flag & 0x4) && !(F.flag & 0x400);
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

// ---- read selection, depth and packing where the records are (include/tredgpu.h section 5) -----------------------------------
// BamParser.parse's selection and BamDepth's pile-up sum as bamread.cpp's scan_impl restates them (tredparse/bam_parser.py:
// 199-243, 404-411), over the record list the chain and parse kernels made for the pair walk: one wavefront per locus goes
// through the region's parsed fields 64 records at a time -- a ballot and a prefix popcount keep the file's order --, then
// appends the alternative regions' hits (their virtual offsets, turned back into places of `out`), region by region as the
// host does.  What comes out per locus: where each selected record lies, how many there are and what room they take, the
// depth sum.  Anything the pair walk or an alternative region's walk declined is declined here too (the status is passed on:
// the host then scans that sample itself).
constexpr int SEL_CAP = TREDGPU_SELECT_CAP;
enum { SEL_FULL = 8, SEL_LONG_READ = 9 };

__device__ inline long long wave_sum64(long long x) {
    for (int d = 32; d >= 1; d >>= 1) x += (long long)__shfl_xor((unsigned long long)x, d, LANES);
    return x;
}
__device__ inline int wave_sum32(int x) {
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, LANES);
    return x;
}
__device__ inline int wave_max32(int x) {
    for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d, LANES));
    return x;
}

__global__ void __launch_bounds__(LANES) select_kernel(WalkView v, const tredgpu_walk_task* tasks, const tredgpu_select_task* sel,
                                                       const int64_t* rec_base, const WalkRec* recs_all, const WalkFields* fields_all,
                                                       const WalkChained* chained, const tredgpu_walk_result* results,
                                                       const tredgpu_walk_task* alt_tasks, const tredgpu_alt_result* alt_results, int n_alt_tasks,
                                                       int64_t* sel_list, tredgpu_select_result* out) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const tredgpu_walk_task T = tasks[t];
    const tredgpu_select_task S = sel[t];
    const WalkChained C = chained[t];
    const uint8_t* o = v.out;
    int64_t* list = sel_list + (size_t)t * SEL_CAP;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    int status = results[t].status;
    int cnt = 0;
    long long depth = 0;                               // (per-lane partial sums: added up at the end)
    int words = 0, s4 = 0, nmb = 0, maxlen = 0;
    auto take = [&](int64_t a0, int idx) {
        const int64_t r = a0 + 4;
        const int l_name = (int)g_u8(o, r + 8), L = (int)g_u32(o, r + 16);
        if (idx < SEL_CAP) list[idx] = a0;
        words += ((L + 15) >> 4) + ((L + 31) >> 5);
        s4 += (L + 1) >> 1;
        nmb += max(l_name - 1, 0);
        maxlen = max(maxlen, L);
    };
    if (status == WALK_OK) {
        const WalkRec* recs = recs_all + rec_base[t];
        const WalkFields* fields = fields_all + rec_base[t];
        const int n = C.n;
        for (int b = 0; b * LANES < n; ++b) {
            const int r = b * LANES + lane;
            const bool mine = r < n;
            WalkFields F = {};
            if (mine) F = fields[r];
            const bool off_region = F.rtid != T.tid || F.rpos >= T.end;
            const int64_t e = (F.rend < 0 || F.rend <= F.rpos) ? (int64_t)F.rpos + 1 : (int64_t)F.rend;
            const bool inwin = mine && !off_region && e > T.start && F.rpos < T.win_hi && e > T.win_lo;   // the window's query returns it
            if (inwin && !(F.flag & (0x4 | 0x100 | 0x200 | 0x400)) && F.rend >= 0) depth += (long long)F.rend - F.rpos;
            const bool pick = inwin && S.n_alt >= 0 && ((F.flag & 0x4) != 0 || (F.rpos >= S.pos_lo && F.rpos <= S.pos_hi));
            const uint64_t m = __ballot(pick);
            if (pick) take(recs[r].a0, cnt + __popcll(m & below));
            cnt += __popcll(m);
        }
        // the alternative regions' hits, region by region (scan_impl: pool_walked), every lane one hit
        for (int k = 0; k < S.n_alt && status == WALK_OK; ++k) {
            const int q = S.alt_first + k;
            if (q < 0 || q >= n_alt_tasks) { status = WALK_NOT_PLANNED; break; }
            const tredgpu_walk_task AT = alt_tasks[q];
            if (AT.n_chunks < 0) continue;             // (its contig is not in this file)
            const tredgpu_alt_result AR = alt_results[q];
            if (AR.status != WALK_OK || AR.n < 0 || AR.n > 6) { status = AR.status != WALK_OK ? AR.status : (int)WALK_POOL_FULL; break; }
            bool lost = false;
            if (lane < AR.n) {
                uint64_t at = 0;
                for (int m = 0; m < 6; ++m) if (m == lane) at = AR.vbeg[m];
                const int64_t coff = (int64_t)(at >> 16);
                int lo = AT.block_first, hi = AT.block_end;          // the block of the sample's that starts at coff
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (v.bcoff[mid] < coff) lo = mid + 1; else hi = mid;
                }
                if (lo >= AT.block_end || v.bcoff[lo] != coff) lost = true;
                else take(v.ooff[lo] + (int64_t)(at & 0xFFFFu), cnt + lane);
            }
            if (__ballot(lost) != 0) { status = WALK_NOT_PLANNED; break; }
            cnt += AR.n;
        }
    }
    depth = wave_sum64(depth);
    words = wave_sum32(words); s4 = wave_sum32(s4); nmb = wave_sum32(nmb); maxlen = wave_max32(maxlen);
    if (status == WALK_OK && cnt > SEL_CAP) status = SEL_FULL;
    if (status == WALK_OK && maxlen > TREDGPU_MAX_READ_LEN) status = SEL_LONG_READ;
    if (lane == 0) {
        tredgpu_select_result R = {};
        R.status = status;
        if (status == WALK_OK) { R.n_reads = cnt; R.n_words = words; R.seq4_bytes = s4; R.name_bytes = nmb; R.max_len = maxlen; R.depth_sum = depth; }
        out[t] = R;
    }
}

// The selected reads of a unit into libtredgpu's read layout (tredgpu_pack_reads: ceil(L/16) words of 2-bit codes, then
// ceil(L/32) words of N flags; bamread.cpp pool_read) and, for the writers, their 4-bit sequences and names one after the
// other.  A workgroup of four wavefronts per unit: first every read's place (a block-wide running sum over the reads' sizes,
// from the unit's bases the caller summed up from the select results), then a wavefront per read: a lane per output word.
__global__ void __launch_bounds__(256) pack_selected_kernel(const uint8_t* __restrict__ o, const int64_t* __restrict__ sel_list,
                                                            const int32_t* __restrict__ unit_task, const int32_t* __restrict__ unit_read_off,
                                                            const int64_t* __restrict__ unit_word_off, const int64_t* __restrict__ unit_seq4_off,
                                                            const int64_t* __restrict__ unit_name_off, int u0, uint32_t* packed, int64_t* read_off,
                                                            int32_t* read_len, uint8_t* seq4, int64_t* seq4_off, uint8_t* names, int64_t* name_off) {
    __shared__ int wsum[3][4];
    const int u = u0 + blockIdx.x, tid = threadIdx.x, lane = tid & (LANES - 1), w = tid / LANES;
    const int64_t* list = sel_list + (size_t)unit_task[u] * SEL_CAP;
    const int r0 = unit_read_off[u], n = unit_read_off[u + 1] - r0;
    long long base_w = unit_word_off[u], base_s = unit_seq4_off[u], base_n = unit_name_off[u];
    for (int c = 0; c < n; c += 256) {
        const int i = c + tid;
        int lw = 0, ls = 0, ln = 0, L = 0;
        if (i < n) {
            const int64_t r = list[i] + 4;
            const int l_name = (int)g_u8(o, r + 8);
            L = (int)g_u32(o, r + 16);
            lw = ((L + 15) >> 4) + ((L + 31) >> 5);
            ls = (L + 1) >> 1;
            ln = max(l_name - 1, 0);
        }
        const int iw = wave_incl_scan(lw), is = wave_incl_scan(ls), in = wave_incl_scan(ln);
        if (lane == LANES - 1) { wsum[0][w] = iw; wsum[1][w] = is; wsum[2][w] = in; }
        __syncthreads();
        int pw = 0, ps = 0, pn = 0, tw = 0, ts = 0, tn = 0;
        for (int k = 0; k < 4; ++k) {
            if (k < w) { pw += wsum[0][k]; ps += wsum[1][k]; pn += wsum[2][k]; }
            tw += wsum[0][k]; ts += wsum[1][k]; tn += wsum[2][k];
        }
        if (i < n) {
            read_off[r0 + i] = base_w + pw + iw - lw;
            read_len[r0 + i] = L;
            seq4_off[r0 + i] = base_s + ps + is - ls;
            name_off[r0 + i] = base_n + pn + in - ln;
        }
        base_w += tw; base_s += ts; base_n += tn;
        __syncthreads();
    }
    // (the entry behind the unit's last read is the next unit's first -- the same number -- or the arrays' end)
    if (tid == 0) { read_off[r0 + n] = unit_word_off[u + 1]; seq4_off[r0 + n] = unit_seq4_off[u + 1]; name_off[r0 + n] = unit_name_off[u + 1]; }
    __syncthreads();
    for (int i = w; i < n; i += 4) {
        const int64_t r = list[i] + 4;
        const int l_name = (int)g_u8(o, r + 8), n_cigar = (int)g_u16(o, r + 12), L = (int)g_u32(o, r + 16);
        const int64_t seq = r + 32 + l_name + 4 * (int64_t)n_cigar;
        const int nb = (L + 15) >> 4, nm = (L + 31) >> 5;
        uint32_t* rec = packed + read_off[r0 + i];
        // "=ACMGRSVTWYHKDBN": A C G T are the codes 0..3, every other letter is an N (code 4)
        constexpr unsigned long long CODE = 0x4444444344424104ull;
        for (int j = lane; j < nb + nm; j += LANES) {
            uint32_t word = 0;
            if (j < nb) {
                for (int q = 0; q < 16; ++q) {
                    const int b = 16 * j + q;
                    if (b < L) {
                        const uint32_t byte = g_u8(o, seq + (b >> 1));
                        const uint32_t code = (uint32_t)(CODE >> (4 * ((b & 1) ? (byte & 15u) : (byte >> 4)))) & 15u;
                        if (code < 4) word |= code << (2 * q);
                    }
                }
            } else {
                for (int q = 0; q < 32; ++q) {
                    const int b = 32 * (j - nb) + q;
                    if (b < L) {
                        const uint32_t byte = g_u8(o, seq + (b >> 1));
                        const uint32_t code = (uint32_t)(CODE >> (4 * ((b & 1) ? (byte & 15u) : (byte >> 4)))) & 15u;
                        if (code == 4) word |= 1u << q;
                    }
                }
            }
            rec[j] = word;
        }
        uint8_t* s4 = seq4 + seq4_off[r0 + i];
        for (int j = lane; j < ((L + 1) >> 1); j += LANES) s4[j] = (uint8_t)g_u8(o, seq + j);
        uint8_t* nm_out = names + name_off[r0 + i];
        for (int j = lane; j < l_name - 1; j += LANES) nm_out[j] = (uint8_t)g_u8(o, r + 32 + j);
    }
}

}  // namespace

namespace tredgpu_front {
hipError_t launch_walk_chain_par(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, const int64_t* rec_base,
                                 WalkRec* recs, WalkChained* chained, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return hipSuccess;
    walk_chain_par_kernel<<<(unsigned)n_tasks, LANES, 0, st>>>(v, tasks, chunks, rec_base, recs, chained);
    return hipGetLastError();
}
hipError_t launch_walk_chain(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, const int64_t* rec_base,
                             WalkRec* recs, WalkChained* chained, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return hipSuccess;
    walk_chain_kernel<<<(unsigned)n_tasks, LANES, 0, st>>>(v, tasks, chunks, rec_base, recs, chained);
    return hipGetLastError();
}
hipError_t launch_walk_parse(const WalkView& v, int n_tasks, const int64_t* rec_base, const WalkRec* recs, const WalkChained* chained,
                             WalkFields* fields, size_t total_recs, hipStream_t st) {
    if (total_recs == 0) return hipSuccess;
    walk_parse_kernel<<<(unsigned)((total_recs + 255) / 256), 256, 0, st>>>(v, n_tasks, rec_base, recs, chained, fields);
    return hipGetLastError();
}
hipError_t allow_pair_walk_lds() {
    return hipFuncSetAttribute((const void*)pair_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds_bytes(WALK_PAIR_CAP));
}
hipError_t launch_pair_walk(const WalkView& v, const tredgpu_walk_task* tasks, const int64_t* rec_base, const WalkRec* recs, const WalkFields* fields,
                            const WalkChained* chained, tredgpu_walk_result* results, WalkPair* pairs, int32_t* gpool, int64_t cap_g, int32_t* tpool,
                            int64_t cap_t, unsigned long long* counters, int table_cap, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return hipSuccess;
    pair_walk_kernel<<<(unsigned)n_tasks, PW_THREADS, walk_lds_bytes(table_cap), st>>>(v, tasks, rec_base, recs, fields, chained, results, pairs, gpool,
                                                                                        cap_g, tpool, cap_t, counters, table_cap);
    return hipGetLastError();
}
hipError_t launch_alt_walk(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, tredgpu_alt_result* results,
                           uint8_t* need, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return hipSuccess;
    alt_walk_kernel<<<(unsigned)n_tasks, LANES, WALK_WINDOW, st>>>(v, tasks, chunks, results, need);
    return hipGetLastError();
}
hipError_t launch_fetch_gather(const uint8_t* out, uint8_t* host, const FetchPiece* pieces, size_t n_pieces, hipStream_t st) {
    if (n_pieces == 0) return hipSuccess;
    fetch_gather_kernel<<<(unsigned)n_pieces, 256, 0, st>>>(out, host, pieces);
    return hipGetLastError();
}
hipError_t launch_select(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_select_task* sel, const int64_t* rec_base,
                         const WalkRec* recs, const WalkFields* fields, const WalkChained* chained, const tredgpu_walk_result* results,
                         const tredgpu_walk_task* alt_tasks, const tredgpu_alt_result* alt_results, int n_alt_tasks, int64_t* sel_list,
                         tredgpu_select_result* out, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return hipSuccess;
    select_kernel<<<(unsigned)n_tasks, LANES, 0, st>>>(v, tasks, sel, rec_base, recs, fields, chained, results, alt_tasks, alt_results, n_alt_tasks,
                                                       sel_list, out);
    return hipGetLastError();
}
hipError_t launch_pack_selected(const uint8_t* out, const int64_t* sel_list, const int32_t* unit_task, const int32_t* unit_read_off,
                                const int64_t* unit_word_off, const int64_t* unit_seq4_off, const int64_t* unit_name_off, int u0, int n_units,
                                uint32_t* packed, int64_t* read_off, int32_t* read_len, uint8_t* seq4, int64_t* seq4_off, uint8_t* names,
                                int64_t* name_off, hipStream_t st) {
    if (n_units <= 0) return hipSuccess;
    pack_selected_kernel<<<(unsigned)n_units, 256, 0, st>>>(out, sel_list, unit_task, unit_read_off, unit_word_off, unit_seq4_off, unit_name_off, u0,
                                                            packed, read_off, read_len, seq4, seq4_off, names, name_off);
    return hipGetLastError();
}
}  // namespace tredgpu_front
