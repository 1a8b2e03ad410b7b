// sw_ladder.hip -- template-ladder Smith-Waterman + read tagging for gfx950 (MI355X, CDNA4).
//
// Replaces, for a whole batch of reads in one launch:
//   /root/reference/src/ssw.c:780-871        ssw_align (forward pass :123-345/:371-547, reverse pass :839-851)
//   /root/reference/src/ssw_wrap.py:177-227  Aligner.align incl. the score / length filter :214-220
//   /root/reference/tredparse/bam_parser.py:123-182  _parseReadSW (hangs :102-121, tags :139-168, arg-max :174)
//
// Design (not a translation of the striped SSE2 code):
//  * one wavefront = four reads of one ladder (any units), 16 lanes per read, R consecutive read
//    rows per lane (R = ceil(maxlen/16)); the DP column lives in registers, lanes talk through DPP
//    row shifts only (no barriers), so a DPP row (16 lanes) is exactly one alignment.
//  * the vertical-gap term F is an exclusive max-plus prefix scan over the 16 lanes (4 DPP steps)
//    instead of Farrar's data-dependent lazy-F loop.
//  * shared-prefix ladder: templates prefix+repeat*u+suffix (u=1..max_units) share the trunk
//    prefix+repeat*max_units, swept once; a forward column depends only on the columns to its left.
//  * suffix continuation vectors: what an alignment can still gain in the |suffix| columns after leaving
//    the trunk at read row i does not depend on u; one reversed alignment of the read against the
//    reversed suffix per strand gives it for every row, and a template's result is the trunk state at
//    its end column combined with those vectors (see the block comment before sweep_column_free).
//  * begin coordinates without the reference's reverse pass: every DP value is one int32
//    score<<18 | start_col<<9 | start_row; integer max then picks (score, largest start column,
//    largest start row), which is what the reverse pass reports (first column walking left whose
//    max equals the score, smallest reversed row).  End coordinates use the key
//    score<<18 | (511-col)<<9 | (511-row): its max is "first column reaching the max, smallest row".
//    Both rules are validated against the compiled reference in oracle/ladder_model.c's tests.
//  * the kernel is bound by VALU issue slots, LDS is idle: what can be left to the LDS pipe is.  A lane that sees a
//    new best cell parks its R row values in LDS and leaves a note (resolve_best works out row and start payload
//    only at template ends that are really combined); the (key, start) maximum of a template end's 2R candidates
//    is taken by 64-bit LDS max atomics on a slot of the lane's own.  No lane reads another lane's slots.
#include "tredgpu_internal.h"

namespace tredgpu {
namespace {

constexpr int KSH = 18;
constexpr int KONE = 1 << KSH;
constexpr int PAYMASK = KONE - 1;
constexpr int NEG = -(1 << 30);
constexpr int PADNEG = -64 * KONE;

template <int CTRL>
__device__ __forceinline__ int dpp_row_shr(int old, int x) {
    // row_shr:n inside a 16-lane DPP row; lanes without a source keep `old`
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, 0xF, false);
}

// v_add3_u32 with the wave-uniform column term in an SGPR.  Written as asm so that the compiler cannot
// re-associate max(diag + S, fresh + S) into max(diag, fresh) + S (one more VALU op per cell).
__device__ __forceinline__ int add3_vsv(int v0, int s1, int v2) {
    int d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(v0), "s"(s1), "v"(v2));
    return d;
}
// v_add_u32 of a wave-uniform term (SGPR) and a lane value, opaque to the optimiser: written out in C the compiler
// hoists "row part + r * stride" for every row out of the column loop -- R more live registers, i.e. scratch.
__device__ __forceinline__ int add_sv(int s0, int v1) {
    int d;
    asm("v_add_u32 %0, %1, %2" : "=v"(d) : "s"(s0), "v"(v1));
    return d;
}
__device__ __forceinline__ int max3(int a, int b, int c) { return max(max(a, b), c); }

// inclusive max scan over the 16 lanes of a DPP row (old = x: lanes without a source keep x)
__device__ __forceinline__ int row_scan_max(int x) {
    x = max(x, dpp_row_shr<0x111>(x, x));
    x = max(x, dpp_row_shr<0x112>(x, x));
    x = max(x, dpp_row_shr<0x114>(x, x));
    x = max(x, dpp_row_shr<0x118>(x, x));
    return x;
}

template <int CTRL>
__device__ __forceinline__ void pair_step(int& k, int& s) {
    const int tk = dpp_row_shr<CTRL>(k, k);
    const int ts = dpp_row_shr<CTRL>(s, s);
    const bool c = tk > k;
    k = c ? tk : k;
    s = c ? ts : s;
}

// Per-lane constants of one read ("job").  All DP values live in the anti-diagonal-scaled domain
//   X~[i][c] = X[i][c] + (i + c) * ge * K
// in which the three recurrences lose their per-cell gap-extension subtractions:
//   F~[i]   = max(F~[i-1], H~[i-1] - c0)          c0 = (go - ge) * K
//   E~[c+1] = max(E~[c],   H~[c]   - c0)
//   H~t     = max(D~ + s + 2geK, fresh~ + s, E~)
// so F~ inside a lane is a plain prefix max and across lanes a plain DPP max scan.
template <int R>
struct Rows {
    int S[4][R];  // (score vs template letter 0..3) * K + 2*ge*K; PADNEG for padding rows
    int rowc0;    // i0 + i0*ge*K for the lane's first row i0: the row part of a fresh start in the scaled
                  // domain; row r adds r*(1 + ge*K), folded into the wave-uniform column term (an SGPR per row)
};

struct Track {
    int bestkey;    // score<<18 | (511-col)<<9 | (511-row), true (unscaled) score; < 0: see resolve_best
    int beststart;  // packed value of that cell (start col/row in the low 18 bits)
    int ceil;       // best score << 18 | PAYMASK: what a cell must exceed to be a new best
};

// inclusive max scan over the 16 lanes of a DPP row fused with the exclusive shift, as one asm block:
// v_max_i32_dpp reads its own and the neighbour's value (lanes without a source are disabled and keep
// theirs); s_nop 1 covers the VALU-write -> DPP-read hazard.  in: x = lane value; out: fin = max over
// the lanes above (NEG for the first lane of the row).
__device__ __forceinline__ int row_excl_scan_max(int x) {
    constexpr int IMIN = -2147483647 - 1;
    x = max(x, dpp_row_shr<0x111>(IMIN, x));
    x = max(x, dpp_row_shr<0x112>(IMIN, x));
    x = max(x, dpp_row_shr<0x114>(IMIN, x));
    x = max(x, dpp_row_shr<0x118>(IMIN, x));
    return dpp_row_shr<0x111>(NEG, x);
}

// max over the 16 lanes of a DPP row, delivered to every lane of the row: four rotate-and-max steps
__device__ __forceinline__ int row_max_all(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xF, 0xF, false));   // row_ror:1
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xF, 0xF, false));   // row_ror:2
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xF, 0xF, false));   // row_ror:4
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));   // row_ror:8
    return x;
}

// One DP column for all rows of the four alignments in this wave; LET = template letter (4 = N).
//   s_fresh = (col<<9) + col*geK - 2*geK   (wave-uniform)      s_scale = col*geK
// The local-alignment floor.  Z(i, c) = "the empty alignment whose first cell will be (i+1, c+1)": score 0, start
// payload (c+1) << 9 | (i+1), in the scale of cell (i, c).  Every stored H is max(H, Z), so that the diagonal step
// H[i][c] + S into (i+1, c+1) is at once "extend" and "start a new alignment here" -- the fresh-start term costs
// one two-operand add (the row part is a register, the column part a scalar) inside the max3 that is there anyway,
// instead of a three-operand add of its own.  A path through a cell of score 0 or less never beats the alignment
// that starts after it (same score or more, later start), so the maxima -- scores, end cells and the start-cell
// tie rule -- are those of the recurrences without the floor (checked by tools/fuzz_parity.py against ssw.c).
//   s_z(col) = ((col + 1) << 9) + 1 + col*geK   (wave-uniform column part of Z)
__device__ __forceinline__ int z_col(int col, int geK) { return ((col + 1) << 9) + 1 + col * geK; }

// state in front of column 0: H = Z(i, -1), no horizontal gap open
template <int R>
__device__ __forceinline__ void column_start(const Rows<R>& J, int (&H)[R], int (&E)[R], int geK) {
    const int zc = z_col(-1, geK);
#pragma unroll
    for (int r = 0; r < R; ++r) { H[r] = J.rowc0 + r * (1 + geK) + zc; E[r] = NEG; }
}

template <int R>
__device__ __forceinline__ void sweep_column(const Rows<R>& J, const int (&SL)[R], int (&H)[R], int (&E)[R], Track& T,
                                             int col, int row0, int s_z, int s_scale, int geK,
                                             int c0, int row0g, int* rowbuf) {
    // last row of the lane above, previous column; above the first row lies Z(-1, col-1) = (col << 9) + (col-2)*geK
    const int hup = __builtin_amdgcn_update_dpp((col << 9) + s_scale - 2 * geK, H[R - 1], 0x111, 0xF, 0xF, false);
    int ht[R], pl[R];
    int diag = hup;
    int run = NEG;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int S = SL[r];
        const int t1 = diag + S;                          // extend the alignment ending at (row-1, col-1), or start here
        const int z = add_sv(s_z + r * (1 + geK), J.rowc0);   // Z(row, col): nothing aligned yet
        const int v = max3(t1, z, E[r]);
        diag = H[r];
        ht[r] = v;
        pl[r] = run;                                      // F~ from this lane's rows above
        const int q = v - c0;
        run = max(run, q);
        // E~ for the next column.  Fed from H without the vertical-gap term: a vertical gap followed by
        // a horizontal one has an equal-score twin (horizontal, then vertical) with the same end points
        // that the F recurrence does admit, so nothing is lost (ssw.c:238 makes the same choice).
        E[r] = max(E[r], q);
    }
    // exclusive max scan across the 16 lanes: F~ entering this lane from the lanes above
    const int fin = row_excl_scan_max(run);
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = max3(ht[r], pl[r], fin);
    // running best: only columns in which some lane could beat its best take the exact path.
    // run + c0 = max over this lane's rows of H~ without the vertical-gap term (a best cell never ends
    // in a gap); minus (row0 + col)*geK it over-estimates every row's true value by <= (R-1)*geK.
    // Two segments: rows below MID over-estimate by < MID*ge, the others by <= (R-1-MID)*ge (pl[MID] is the maximum of
    // the rows above MID).  With one segment the slack was (R-1)*ge = 18 at the default scoring -- more than the 11 a
    // one-period gap costs, so on a periodic trunk nearly every lane was "within reach" in nearly every column.
    constexpr int MID = R / 2;
    const int lane_scale = row0g + s_scale;
    const bool trig = max(pl[MID], run - MID * geK) + c0 - lane_scale > T.ceil;
    if (__builtin_amdgcn_ballot_w64(trig) != 0) {
        // exact: does some row of this lane score more than the lane's best?  (a later column never wins a tie)
        int tr[R];
        int m0 = NEG;
        int rs = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            tr[r] = H[r] - rs;  // still carries the lane's scale, a multiple of K
            rs += geK;
            m0 = max(m0, tr[r]);
        }
        const int top = m0 - lane_scale;
        if (top > T.ceil) {
            // A new best cell somewhere in this lane's rows.  Which row, and that row's start payload, are only needed
            // when a template end is really combined (one column in nine), and by then a later column has usually
            // replaced this one: park the rows in LDS (stores cost no VALU issue slot, the kernel's bound) and leave a
            // note -- bestkey = -1 - column, beststart = the lane's best packed value (its score is what counts).
#pragma unroll
            for (int r = 0; r < R; ++r) rowbuf[r * 64] = tr[r];
            T.bestkey = -1 - col;
            T.beststart = top;
            T.ceil = top | PAYMASK;
        }
    }
}

// Turns the note left by sweep_column into the lane's best cell: the row is the maximum of score<<18 | (R-1-r) over
// the parked rows (smallest row among equal scores), the start payload that row's packed value.  Wave-uniform call.
template <int R>
__device__ __forceinline__ void resolve_best(Track& T, int row0, int row0g, int geK, const int* rowbuf) {
    const bool pend = T.bestkey < 0;
    if (__builtin_amdgcn_ballot_w64(pend) == 0) return;
    const int pcol = -1 - T.bestkey;     // (lanes without a note compute on stale rows and keep what they have)
    int m = NEG;
#pragma unroll
    for (int r = 0; r < R; ++r) m = max(m, (rowbuf[r * 64] & ~PAYMASK) | (R - 1 - r));
    static_assert(R <= 32, "row index in five payload bits");
    const int st = rowbuf[(R - 1 - (m & 31)) * 64];
    const int lane_scale = row0g + pcol * geK;
    // -> score<<18 | (511-col)<<9 | (511-row)
    const int cand = m - lane_scale + (((511 - pcol) << 9) + (511 - (R - 1)) - row0);
    T.bestkey = pend ? cand : T.bestkey;
    T.beststart = pend ? st - lane_scale : T.beststart;
}

// The profile row values of one template letter (wave-uniform): R register moves behind a scalar switch.
template <int R>
__device__ __forceinline__ void pick_profile(const Rows<R>& J, int letter, int geK, int (&S)[R]) {
    switch (letter) {
#define TREDGPU_PICK(K)                              \
    case K:                                          \
        _Pragma("unroll") for (int r = 0; r < R; ++r) S[r] = J.S[K][r]; \
        break;
        TREDGPU_PICK(0)
        TREDGPU_PICK(1)
        TREDGPU_PICK(2)
        TREDGPU_PICK(3)
#undef TREDGPU_PICK
        default:   // N column: 0 against every real row
#pragma unroll
            for (int r = 0; r < R; ++r) S[r] = J.S[0][r] < (PADNEG >> 1) ? PADNEG : 2 * geK;
            break;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(S[r]));   // (keeps the column body from being cloned into the cases)
}

template <int R>
__device__ __forceinline__ void sweep_at(const Rows<R>& J, const int (&S)[R], int (&H)[R], int (&E)[R], Track& T,
                                         int col, int row0, int geK, int c0, int row0g, int* rowbuf) {
    const int s_scale = col * geK;
    sweep_column<R>(J, S, H, E, T, col, row0, z_col(col, geK), s_scale, geK, c0, row0g, rowbuf);
}

// One column whose letter is only known at run time (prefix columns, ladders of other periods, the dump variant's
// suffix sweep).  Only the choice of the profile row values sits in a switch; the column itself is ONE body (with
// the whole column inlined five times behind the switch the compiler reconciled the five copies' register
// allocations with 23 moves at the head of every column).
template <int R>
__device__ __forceinline__ void sweep_letter(int letter, const Rows<R>& J, int (&H)[R], int (&E)[R], Track& T,
                                             int col, int row0, int geK, int c0, int row0g, int* rowbuf) {
    int S[R];
    pick_profile<R>(J, letter, geK, S);
    sweep_at<R>(J, S, H, E, T, col, row0, geK, c0, row0g, rowbuf);
}

// Letters are packed 8 per 32-bit word (4 bits each).  A strand's trunk (<= 64 words) and suffix words
// are loaded once into one VGPR each, word k in lane k, and fetched per column with v_readlane (no
// memory access in the column loop).
__device__ __forceinline__ int letter_from(int words_vgpr, int idx) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(words_vgpr, idx >> 3);
    return (int)((w >> ((idx & 7) * 4)) & 7u);
}

// ---------------------------------------------------------------------------------------------------------
// Suffix continuation.  Every template prefix + repeat*u + suffix ends with the same |suffix| columns, so what
// an alignment can still gain after leaving the trunk at read row i does not depend on u.  One reversed
// alignment of the read against the reversed suffix (once per read and strand, |suffix| columns) yields, for
// every row, the weight and the end cell of the best continuation
//     WH[i]: the alignment leaves trunk cell (i, c) by a match at (i+1, c+1)
//     WE[i]: it leaves inside a horizontal gap (trunk E[i][c+1], next match in row i+1)
// and a template's result is max(trunk best, max_i H[i][c] + WX[i], max_i E[i][c+1] + WE[i], best alignment
// inside the suffix alone), WX = max(WH, WE - (go - ge)): the horizontal gap that opens from the trunk cell itself -- about one column's worth of work per template instead of |suffix| columns.
// Packing makes the tie rules carry over: the reversed alignment's "start" payload (largest reversed column,
// largest reversed row) is the forward end cell with the smallest column, then the smallest row; the start
// coordinates travel in the trunk values' payload.  The reversed pass uses the unrestricted recurrences (E fed
// from H including the vertical-gap term) and the junction admits H - go: every Gotoh path then has an
// equal-score representative that is restricted (no vertical-then-horizontal gap) on the trunk side only.
// tools/proto_continuation.py checks the formulation against the CPU oracle in plain integers.
constexpr int NEGH = -(1 << 29);

template <int R, int LET>
__device__ __forceinline__ void sweep_column_free(const Rows<R>& J, int (&H)[R], int (&E)[R], int s_fresh,
                                                  int geK, int c0) {
    const int hup = dpp_row_shr<0x111>(NEG, H[R - 1]);
    int ht[R], pl[R];
    int diag = hup;
    int run = NEG;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int S;
        if (LET < 4) S = J.S[LET][r];
        else S = J.S[0][r] < (PADNEG >> 1) ? PADNEG : 2 * geK;
        const int t1 = diag + S;
        const int t2 = add3_vsv(J.rowc0, s_fresh + r * (1 + geK), S);
        const int v = max3(t1, t2, E[r]);
        diag = H[r];
        ht[r] = v;
        pl[r] = run;
        run = max(run, v - c0);
    }
    const int fin = row_excl_scan_max(run);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        H[r] = max3(ht[r], pl[r], fin);
        E[r] = max(E[r], H[r] - c0);   // unrestricted: a horizontal gap may follow a vertical one
    }
}

// last reversed column: only the cells entered by a match (or started there) are needed
template <int R, int LET>
__device__ __forceinline__ void match_column(const Rows<R>& J, const int (&H)[R], int (&D)[R], int s_fresh, int geK) {
    int diag = dpp_row_shr<0x111>(NEG, H[R - 1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int S;
        if (LET < 4) S = J.S[LET][r];
        else S = J.S[0][r] < (PADNEG >> 1) ? PADNEG : 2 * geK;
        D[r] = max(diag + S, add3_vsv(J.rowc0, s_fresh + r * (1 + geK), S));
        diag = H[r];
    }
}

__device__ __forceinline__ int row_mirror(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); }

// reversed position x of this lane's register r  ->  forward row (i + 1) = 16R - 1 - x: the vector indexed by
// forward row i is the mirrored one moved up by one row (NEGH for the last row: nothing left to match)
template <int R>
__device__ __forceinline__ void mirror_up(const int (&X)[R], int add, int* out, int stride) {
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) out[r * stride] = max(row_mirror(X[R - 2 - r]) + add, NEGH);
    const int m0 = row_mirror(X[R - 1]);
    const int nx = __builtin_amdgcn_update_dpp(NEGH, m0, 0x101, 0xF, 0xF, false);  // row_shl:1
    out[(R - 1) * stride] = max(nx + add, NEGH);
}

template <int CTRL>
__device__ __forceinline__ void pair_step_lex(int& k, int& s) {
    const int tk = dpp_row_shr<CTRL>(k, k);
    const int ts = dpp_row_shr<CTRL>(s, s);
    const bool c = tk > k || (tk == k && ts > s);
    k = c ? tk : k;
    s = c ? ts : s;
}

// Codes of R consecutive read rows i_lo .. i_lo+R-1 (R <= 24): they span at most three words of the 2-bit codes and
// two words of the N mask (tredgpu.h read packing), so five loads fetch them all.  code2: 2 bits per row (32 rows'
// worth); nmask: bit k = row i_lo+k is N.  Rows >= L hold garbage (the caller pads them).
template <int R>
__device__ __forceinline__ void load_rows(const SwArgs& a, int64_t off, int L, int i_lo, uint64_t& code2, uint32_t& nmask) {
    // (R = 32: a lane's rows start at a multiple of 32 -- two code words and one mask word, nothing to shift)
    static_assert(R <= 24 || R == 32, "three code words cover 33 rows from any offset, the mask window 33");
    const int nb = (L + 15) >> 4, nm = (L + 31) >> 5;
    const int wl = max(nb - 1, 0), ml = max(nm - 1, 0);
    const int wi = i_lo >> 4, mi = i_lo >> 5;
    const uint32_t w0 = a.packed[off + min(wi, wl)], w1 = a.packed[off + min(wi + 1, wl)];
    const uint32_t m0 = a.packed[off + nb + min(mi, ml)], m1 = a.packed[off + nb + min(mi + 1, ml)];
    const int sh = (i_lo & 15) * 2;
    code2 = (((uint64_t)w1 << 32) | w0) >> sh;
    if (R > 16) {   // rows 17.. of the window sit in the third word
        const uint32_t w2 = a.packed[off + min(wi + 2, wl)];
        code2 |= sh ? (uint64_t)w2 << (64 - sh) : 0;
    }
    nmask = (uint32_t)((((uint64_t)m1 << 32) | m0) >> (i_lo & 31));
}

// Match/mismatch profile of this lane's R rows.  reversed: position x of the lane layout holds read row
// 16R-1-x (padding first), the layout of the continuation pass.
template <int R>
__device__ __forceinline__ void build_profile(Rows<R>& J, const SwArgs& a, int64_t off, int L, int row0, bool reversed,
                                              int mK, int xK, int geK) {
    // (the profile is the same for both strands; the empty asm keeps the compiler from hoisting it out of the
    //  strand loop, where the forward and the reversed one would be live together: 2 x 4R registers)
    asm volatile("" : "+v"(L), "+v"(off), "+v"(row0));   // (likewise the addresses and shifts of the four loads)
    J.rowc0 = row0 + row0 * geK;
    const int i_lo = reversed ? 16 * R - R - row0 : row0;
    uint64_t code2;
    uint32_t nmask;
    load_rows<R>(a, off, L, i_lo, code2, nmask);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = reversed ? R - 1 - r : r;
        const int i = i_lo + k;
        const int code = i >= L ? 5 : (((nmask >> k) & 1u) ? 4 : (int)((code2 >> (2 * k)) & 3u));
#pragma unroll
        for (int l = 0; l < 4; ++l)
            J.S[l][r] = code == 5 ? PADNEG : (code == 4 ? 0 : (code == l ? mK : xK)) + 2 * geK;
    }
}

// The lane index, recomputed on the spot (two instructions) instead of held in a register from the kernel's entry
// to its last line: with 64-thread workgroups threadIdx.x is the lane.  Used by the cold parts of the kernel so
// that nothing derived from the thread id has to survive the column loop in scratch.
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// GENERIC = false is the production kernel: no per-template dump, and no sweep of the suffix alone (an alignment
// inside an 18-base suffix cannot reach the score filter of 30).  The host picks GENERIC = true when a dump is asked
// for or some registered ladder has a branch long enough to reach the filter on its own (|branch| * match >= 30).
// Waves per SIMD each instantiation is compiled for (make EXTRA="-DSW_WAVES_R10=5" builds a variant; tools/sw_waves_ab.sh).
// Round 6 (profiles/r06_sw_waves_ab.txt): the 112-row kernel at 6 instead of 4 (80 VGPRs, a few spilled: 12.43 -> 12.06 ms
// per 500-sample step at 100 bp, frac 0.306 -> 0.334) and the 256-row kernel at 3 instead of 2 (168 VGPRs, 50 spilled to
// scratch that stays in L1/L2: 19.29 -> 16.74 ms at 250 bp, frac 0.274 -> 0.319) -- the waves hide more of the DPP scan's
// and the LDS notes' latency than the spills cost; the 160-row kernel loses at 5 and 6 (15.98 -> 16.57 / 18.21 ms) and stays at 4.
#ifndef SW_WAVES_R4
#define SW_WAVES_R4 6
#endif
#ifndef SW_WAVES_R7
#define SW_WAVES_R7 6
#endif
#ifndef SW_WAVES_R10
#define SW_WAVES_R10 4
#endif
#ifndef SW_WAVES_R16
#define SW_WAVES_R16 3
#endif
#ifndef SW_WAVES_R20
#define SW_WAVES_R20 2
#endif
#ifndef SW_WAVES_R32
#define SW_WAVES_R32 1
#endif
template <int R, int W, bool GENERIC>
__global__ __launch_bounds__(64, W) void sw_cont_kernel(SwArgs a) {
    // One wavefront per workgroup: quads differ a lot in length (pruning), and a wave slot freed by a short
    // quad is only refilled when a whole new workgroup fits.
    // continuation vectors of the strand being swept: [WH rows | WE rows | suffix-only best: key, start][64 lanes]
    constexpr int PS = 64;
    // + R rows per lane: where the exact path of sweep_column picks the start payload of a new best cell
    // + one 64-bit slot per lane: the (key, start) maximum of a template end
    __shared__ __attribute__((aligned(8))) int wbuf[(3 * R + 2) * PS + 2 * PS];
    const int lane = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x;
    const int nq = *a.n_quads;
    if (q >= nq) return;
    const int q_ladder = __builtin_amdgcn_readfirstlane(a.quads[q].ladder);
    const int q_first = __builtin_amdgcn_readfirstlane(a.quads[q].first);
    const int q_strands = __builtin_amdgcn_readfirstlane(a.quads[q].strands);
    const int q_count = __builtin_amdgcn_readfirstlane(a.quads[q].count);
    const LadderDesc* ld = a.ladders + q_ladder;
    const int period = __builtin_amdgcn_readfirstlane(ld->period);
    const int max_units = __builtin_amdgcn_readfirstlane(ld->max_units);
    const int n_strands = __builtin_amdgcn_readfirstlane(ld->n_strands);

    const int job = lane >> 4, jl = lane & 15;
    const bool valid = job < q_count;
    int L, row0;
    bool too_long;
    {
        const int64_t rd = valid ? (int64_t)a.perm[q_first + job] : 0;
        L = valid ? a.read_len[rd] : 0;
        too_long = L > 16 * R;  // not representable in this instantiation: flagged, not aligned
        if (too_long) L = 0;
        row0 = jl * R;
    }
    // first packed word of this lane's read, looked up again wherever the read's codes are loaded (cold code)
    auto read_base = [&]() -> int64_t {
        const int jb = lane_now() >> 4;
        return jb < q_count ? a.read_off[a.perm[q_first + jb]] : 0;
    };
    const int mK = a.p.match * KONE;
    const int xK = -a.p.mismatch * KONE;
    const int geK = a.p.gap_extend * KONE;
    const int c0 = (a.p.gap_open - a.p.gap_extend) * KONE;
    const int flank = a.p.flank;
    const bool full_dump = GENERIC && a.out_dump != nullptr;  // wave-uniform
    int* const wb = wbuf + lane;
    int* const rowbuf = wb + (2 * R + 2) * PS;

    // REPT cut-off: per-read ceil(L/period) with --useclippedreads, else the ladder's (bam_parser.py:154-155)
    const int mu_rept = a.p.clip ? (L + period - 1) / period : max_units;

    // arg-max so far, one word: (score << 9 | 511 - units) << 3 | tag; -1 = nothing yet.  A candidate must beat
    // it on (score, -units): max(res, key=(score, -units)), first maximal element in db order (bam_parser.py:174)
    int best = -1;
    // wave-uniform work counters, packed so that they cost two scalar registers and never a scratch slot:
    // cnt_cols = trunk columns | continuation columns << 16;  cnt_ends = combined | dropped << 8 | from trunk << 16
    // (a strand has at most 511 columns and 127 templates)
    uint32_t cnt_cols = 0, cnt_ends = 0;

    // Only cells that can survive the score filter are tracked: min_score >= 30 (bam_parser.py:134), so a
    // floor of 29 is exact for tagging; the per-template dump (parity/debug) tracks every positive score.
    // 6-mer filter premise: every run break costs >= 5 matches
    const int kmer_thr = min(a.p.mismatch, a.p.gap_open) >= 5 * a.p.match ? (30 + a.p.match - 1) / a.p.match - 5 : 0;
    const int floor_key = (full_dump ? 0 : 29) << KSH | PAYMASK;

    for (int s = 0; s < n_strands; ++s) {
        if (!((q_strands >> s) & 1)) continue;
        const int trunk_w = __builtin_amdgcn_readfirstlane(ld->trunk_off[s]);
        const int alen = __builtin_amdgcn_readfirstlane(ld->alen[s]);
        const int blen = __builtin_amdgcn_readfirstlane(ld->blen[s]);
        const int ncols = alen + period * max_units;
        // ---- exact strand filter.  An alignment scoring s with m matches has <= (m - s)/P run breaks
        // (P = min(mismatch, gap_open) >= 5*match), so at least m - 5*(breaks+1) >= s/match - 5 of its
        // 6-mers are exact; no template of this strand can reach the score filter (>= 30,
        // bam_parser.py:134) unless that many read 6-mers occur somewhere in the strand's templates.
        // Read windows containing N count as present; a template N scores 0 against every base, so template
        // windows with N stand for all their fillings (tredgpu_set_ladders).  Skipped for the dump (all positive
        // scores wanted) and for scorings outside the bound's premise.
        int kcap = 1 << 20;  // per read: upper bound of any score on this strand
        // (a 6-mer window starting in this lane ends at most 5 bases into the next lane's rows: needs R >= 5;
        //  with fewer rows per lane the per-read classes of read_class_kernel already removed hopeless strands)
        if constexpr (R >= 5 && 2 * R + 10 <= 32) if (!full_dump && kmer_thr > 0 && __builtin_amdgcn_readfirstlane(ld->kmer_ok) != 0) {
            const uint32_t* bm = a.seqw + __builtin_amdgcn_readfirstlane(ld->kmer_off[s]);
            // this lane's rows as 2-bit codes / N-or-padding flags, loaded here, per strand: they and the window
            // indices below depend on the read only, and hoisted out of the strand loop they would sit in 2R + 2
            // registers across the column loop -- i.e. in scratch
            uint32_t pk_s, nk_s;
            {
                int L_s = L, row0_s = row0;
                asm volatile("" : "+v"(L_s), "+v"(row0_s));
                uint64_t code2;
                uint32_t nmask;
                load_rows<R>(a, read_base(), L_s, row0_s, code2, nmask);
                const int n_real = min(max(L_s - row0_s, 0), R);                 // rows of this lane inside the read
                pk_s = (uint32_t)code2 & ((1u << (2 * R)) - 1u);                             // (codes of N / padding rows are never
                nk_s = (nmask | ~((1u << n_real) - 1u)) & ((1u << R) - 1u);        //  looked at: their windows count as present)
            }
            const uint32_t pk_n = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk_s, 0x101, 0xF, 0xF, false);        // row_shl:1
            const uint32_t nk_n = (uint32_t)__builtin_amdgcn_update_dpp(0x3FF, (int)nk_s, 0x101, 0xF, 0xF, false);
            const uint32_t comb = pk_s | ((pk_n & 0x3FFu) << (2 * R));
            const uint32_t ncomb = nk_s | ((nk_n & 0x1Fu) << R);
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t win = (comb >> (2 * r)) & 0xFFFu;
                const bool has_n = ((ncomb >> r) & 0x3Fu) != 0;
                const bool in_read = row0 + r + 5 < L;
                const uint32_t word = bm[win >> 5];
                cnt += in_read && (has_n || ((word >> (win & 31)) & 1u));
            }
            cnt += dpp_row_shr<0x111>(0, cnt);
            cnt += dpp_row_shr<0x112>(0, cnt);
            cnt += dpp_row_shr<0x114>(0, cnt);
            cnt += dpp_row_shr<0x118>(0, cnt);
            kcap = (cnt + 5) * a.p.match;  // lane 15 of each read holds the read's total
            if (__builtin_amdgcn_ballot_w64(valid && jl == 15 && cnt >= kmer_thr) == 0) continue;
        }
        // Strand exit.  No alignment scores more than cap; a template only matters if it passes its score filter
        // and beats the read's arg-max so far, and at equal score the smaller unit count wins: once neither is
        // possible any more for any read of the wave, the rest of this strand changes nothing.  Quads hold reads
        // of one level of the 6-mer count (read_class_kernel), so the four reads get there together.
        const int cap = min(kcap, L * a.p.match);
        // template un can still matter for this read: its score filter max(min(L, T)/2, 30) (bam_parser.py:133-134,
        // growing with un) and the arg-max so far are within reach of cap
        auto still_open = [&](int un) {
            const int min_score = max(min(L, alen + period * un + blen) >> 1, 30);
            return max(min_score, (best >> 12) + 1) <= cap;
        };
        if (!full_dump && __builtin_amdgcn_ballot_w64(valid && jl == 15 && still_open(max_units > 0 ? 1 : 0)) == 0) continue;

        Rows<R> J;
        int H[R], E[R];
        const bool has_suffix = max_units > 0 && blen > 0;
        if (has_suffix) {
            // ---- continuation vectors of this strand (see above) ----
            build_profile<R>(J, a, read_base(), L, row0, true, mK, xK, geK);
            const int branch_w = __builtin_amdgcn_readfirstlane(ld->branch_off[s]);
            const int ln = lane_now();
            const int bw = ln < ((blen + 7) >> 3) ? (int)a.seqw[branch_w + ln] : 0;
            int D[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = NEG; E[r] = NEG; }
            for (int y = 0; y + 1 < blen; ++y) {
                const int s_fresh = (y << 9) + y * geK - 2 * geK;
                switch (letter_from(bw, blen - 1 - y)) {
                    case 0: sweep_column_free<R, 0>(J, H, E, s_fresh, geK, c0); break;
                    case 1: sweep_column_free<R, 1>(J, H, E, s_fresh, geK, c0); break;
                    case 2: sweep_column_free<R, 2>(J, H, E, s_fresh, geK, c0); break;
                    case 3: sweep_column_free<R, 3>(J, H, E, s_fresh, geK, c0); break;
                    default: sweep_column_free<R, 4>(J, H, E, s_fresh, geK, c0); break;
                }
            }
            {
                const int y = blen - 1;
                const int s_fresh = (y << 9) + y * geK - 2 * geK;
                switch (letter_from(bw, 0)) {
                    case 0: match_column<R, 0>(J, H, D, s_fresh, geK); break;
                    case 1: match_column<R, 1>(J, H, D, s_fresh, geK); break;
                    case 2: match_column<R, 2>(J, H, D, s_fresh, geK); break;
                    case 3: match_column<R, 3>(J, H, D, s_fresh, geK); break;
                    default: match_column<R, 4>(J, H, D, s_fresh, geK); break;
                }
            }
            // E holds the horizontal-gap state of the last reversed column (= first suffix column); its opening
            // cost was charged on the reversed side too: + go - ge in the scaled domain
            mirror_up<R>(D, 0, wb, PS);
            mirror_up<R>(E, c0, wb + R * PS, PS);
            // A horizontal gap into the suffix may also open from the trunk cell itself (H - go, incl. cells reached by a
            // vertical gap): that candidate has H's start payload and the key H + (WE - c0) -- so it folds into the
            // H-candidate's vector once per strand, WX = max(WH, WE - c0), instead of a max(E, H - c0) per row and
            // template end.  (c0 is a multiple of K: the end cell in the low bits is untouched.)
#pragma unroll
            for (int r = 0; r < R; ++r) wb[r * PS] = max(wb[r * PS], wb[(R + r) * PS] - c0);
            cnt_cols += (uint32_t)blen << 16;
        }
        build_profile<R>(J, a, read_base(), L, row0, false, mK, xK, geK);
        const int row0g = row0 * geK;
        Track T;
        // best alignment inside the suffix alone (columns relative to the suffix); it cannot reach the score
        // floor unless the suffix is long enough, so normally only the dump needs it
        const bool with_sfx = GENERIC && has_suffix && (full_dump || blen * a.p.match >= 30);
        if (with_sfx) {
            const int branch_w = __builtin_amdgcn_readfirstlane(ld->branch_off[s]);
            const int ln = lane_now();
            const int bw = ln < ((blen + 7) >> 3) ? (int)a.seqw[branch_w + ln] : 0;
            column_start<R>(J, H, E, geK);
            T.bestkey = floor_key; T.beststart = 0; T.ceil = floor_key;
            for (int j = 0; j < blen; ++j) sweep_letter<R>(letter_from(bw, j), J, H, E, T, j, row0, geK, c0, row0g, rowbuf);
            resolve_best<R>(T, row0, row0g, geK, rowbuf);
            int sk = T.bestkey, ss = T.beststart;
            pair_step<0x111>(sk, ss);
            pair_step<0x112>(sk, ss);
            pair_step<0x114>(sk, ss);
            pair_step<0x118>(sk, ss);
            wb[(2 * R) * PS] = sk;
            wb[(2 * R + 1) * PS] = ss;
        }
        const int ln_t = lane_now();
        const int tw = ln_t < ((ncols + 7) >> 3) ? (int)a.seqw[trunk_w + ln_t] : 0;
        column_start<R>(J, H, E, geK);
        T.bestkey = floor_key; T.beststart = 0; T.ceil = floor_key;
        int next_end = max_units > 0 ? alen + period - 1 : alen - 1;
        int u = max_units > 0 ? 1 : 0;
        // ---- what happens when template u ends at column `col` of the trunk; true = leave this strand ----
        auto template_end = [&](int col) -> bool {
            // ---- template u ends here on the trunk ----
            // (Round 1 first tested an upper bound -- trunk best, or column max + |suffix| * match -- and dropped the
            //  template without looking at the continuation vectors when the bound could not matter.  The exact score
            //  from the first combine pass below drops the same templates; testing the bound first cost more on the
            //  templates that survive it than it saved on the others: 20.4 -> 20.0 ms without it.)
            const bool comb = blen > 0;
            const int Tlen = alen + period * u + blen;
            next_end += period;
            int bk, bs;
            if (comb) {
                // candidates entering the suffix: key = (H or E score field) + continuation (score | end cell), and
                // behind it the start payload the candidate carries.  The lane's largest (key, payload) pair is taken
                // by 64-bit max atomics on a slot of the lane's own in LDS: the 2R compare-and-keep steps cost no VALU
                // issue slot (a second pass over the rows to find the winner's payload cost as much as the first).
                long long* const slot = reinterpret_cast<long long*>(wbuf + (3 * R + 2) * PS) + lane_now();
                *slot = (long long)NEG * (1LL << 32);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int sh = (H[r] & ~PAYMASK) + wb[r * PS], se = (E[r] & ~PAYMASK) + wb[(R + r) * PS];
                    atomicMax(slot, (long long)(((unsigned long long)(uint32_t)sh << 32) | (uint32_t)(H[r] & PAYMASK)));
                    atomicMax(slot, (long long)(((unsigned long long)(uint32_t)se << 32) | (uint32_t)(E[r] & PAYMASK)));
                }
                const long long won = *slot;
                int m = (int)(won >> 32);
                const int st = (int)(won & PAYMASK);
                // scaled sum -> true score; reversed start cell -> 511 - end column | 511 - end row
                const int cu = ((511 - col - blen) << 9) + (512 - 16 * R) - (col + 16 * R + blen - 3) * geK;
                m += cu;
                if (!full_dump) {
                    // the template's exact score is known now (T.ceil carries the trunk's best score): drop it unless
                    // it can pass the score filter and beat the arg-max of some read
                    int ks = max(m, T.ceil);
                    ks = max(ks, dpp_row_shr<0x111>(ks, ks));
                    ks = max(ks, dpp_row_shr<0x112>(ks, ks));
                    ks = max(ks, dpp_row_shr<0x114>(ks, ks));
                    ks = max(ks, dpp_row_shr<0x118>(ks, ks));
                    if (with_sfx) ks = max(ks, wb[(2 * R) * PS]);   // (its column shift does not touch the score)
                    const int bestS = best >> 12, bestU = 511 - ((best >> 3) & 511);
                    const int need_score = max(max(min(L, Tlen) >> 1, 30), u >= bestU ? bestS + 1 : bestS);
                    if (__builtin_amdgcn_ballot_w64(valid && jl == 15 && (ks >> KSH) >= need_score) == 0) {
                        cnt_ends += 1u << 8;
                        ++u;
                        if (__builtin_amdgcn_ballot_w64(valid && jl == 15 && still_open(u)) == 0) return true;
                        return false;
                    }
                }
                resolve_best<R>(T, row0, row0g, geK, rowbuf);
                bk = T.bestkey, bs = T.beststart;
                const bool c = m > bk;     // an equal key is impossible: trunk cells end at columns <= col
                ++cnt_ends;
                bk = c ? m : bk;
                bs = c ? st : bs;
                // largest (key, start) of the read's 16 lanes, in every lane: the key's row maximum first, then the
                // largest start payload among the lanes that hold it (payloads are >= 0)
                {
                    const int bkm = row_max_all(bk);
                    bs = row_max_all(bk == bkm ? bs : -1);
                    bk = bkm;
                }
                if (with_sfx) {
                    const int sk = wb[(2 * R) * PS], ss = wb[(2 * R + 1) * PS];
                    const int k3 = sk - ((col + 1) << 9), s3 = ss + ((col + 1) << 9);
                    const bool c3 = sk != floor_key && (k3 > bk || (k3 == bk && s3 > bs));
                    bk = c3 ? k3 : bk;
                    bs = c3 ? s3 : bs;
                }
            } else {
                cnt_ends += 1u << 16;
                resolve_best<R>(T, row0, row0g, geK, rowbuf);
                bk = T.bestkey, bs = T.beststart;
                pair_step<0x111>(bk, bs);
                pair_step<0x112>(bk, bs);
                pair_step<0x114>(bk, bs);
                pair_step<0x118>(bk, bs);
            }
            const int score = bk >> KSH;
            const int ref_end = 511 - ((bk >> 9) & 511), read_end = 511 - (bk & 511);
            const int ref_begin = (bs >> 9) & 511, read_begin = bs & 511;
            const int min_len = min(L, Tlen) >> 1;              // bam_parser.py:133
            const int min_score = max(min_len, 30);             // :134
            const bool pass = score >= min_score && (read_end - read_begin + 1) >= min_len;  // ssw_wrap.py:217
            const int aL = ref_begin, aR = Tlen - ref_end - 1, bL = read_begin, bR = L - read_end - 1;
            const int hang = min(min(aR + bL, aL + bR), min(aL + aR, bL + bR));  // bam_parser.py:113-121
            const bool prefix_read = ref_begin < flank;                           // :139
            const bool suffix_read = ref_end > Tlen - flank - 1;                  // :140
            int tag;
            if (hang >= flank) tag = TREDGPU_TAG_HANG;
            else if (prefix_read) tag = suffix_read ? TREDGPU_TAG_FULL : TREDGPU_TAG_PREF;
            else if (suffix_read) tag = TREDGPU_TAG_POST;
            else if (u >= mu_rept - 1 && u * period <= L) tag = TREDGPU_TAG_REPT;
            else tag = TREDGPU_TAG_NONE;
            if (!pass) tag = TREDGPU_TAG_NONE;
            const int cand = (score << 9 | (511 - u)) << 3;
            best = tag != TREDGPU_TAG_NONE && cand > (best | 7) ? cand | tag : best;
            if (full_dump && valid && jl == 15) {
                const int k = max_units > 0 ? 2 * (u - 1) + s : 0;
                if (k < a.dump_templates) {
                    int16_t* d = a.out_dump + ((int64_t)a.perm[q_first + job] * a.dump_templates + k) * 6;
                    const bool hit = score > 0 && bk != floor_key;
                    d[0] = (int16_t)(hit ? score : 0);
                    d[1] = (int16_t)(hit ? ref_begin : -1);
                    d[2] = (int16_t)(hit ? ref_end : -1);
                    d[3] = (int16_t)(hit ? read_begin : 0);
                    d[4] = (int16_t)(hit ? read_end : 0);
                    d[5] = (int16_t)tag;
                }
            }
            ++u;
            if (!full_dump && __builtin_amdgcn_ballot_w64(valid && jl == 15 && still_open(u)) == 0) return true;
            return false;
        };
        // Steady-state exit.  If every H and E entry of the column at a period boundary c records an alignment that
        // started inside the repeat (start column >= alen), and one period later every entry one that started at
        // alen + period or later, the two columns are shifts of each other: an optimal alignment of the later column
        // moved one period to the left is an alignment of the earlier one and vice versa, so the scores are equal and
        // the recorded (largest) starts differ by exactly the period -- and then every later column is the shift of the
        // column one period before it (same letters, same recurrence).  Every later template end then has the score of
        // this one (the trunk's best does not move, the continuation vectors do not depend on u); the score a template
        // needs never decreases with u; so after a template end that was DROPPED in that state all later ones would be
        // dropped too: the strand is done.  (Typically 5-6 periods after a read's peak: that long a horizontal gap out
        // of the best alignment still beats an alignment that starts afresh inside the repeat.)  Tested only after
        // dropped template ends -- reads before their peak keep improving -- and never for the dump.
        int ok_run = 0;
        auto steady_exit = [&](const uint32_t ends_before) -> bool {
            if (full_dump || ((cnt_ends ^ ends_before) & 0xFF00u) == 0) { ok_run = 0; return false; }
            const uint32_t thr = (uint32_t)(alen + (ok_run ? period : 0)) << 9;
            uint32_t lo = 0x3FFFFu;
#pragma unroll
            for (int r = 0; r < R; ++r) lo = min(lo, min((uint32_t)H[r] & 0x3FE00u, (uint32_t)E[r] & 0x3FE00u));
            if (__builtin_amdgcn_ballot_w64(lo < thr) != 0) { ok_run = 0; return false; }
            if (ok_run) return true;
            ok_run = 1;
            return false;
        };
        if (period == 3 && max_units > 0) {
            // Period-3 ladders (27 of the 30 loci): inside the repeat the column letters cycle through the motif, so
            // the three profile rows are picked once per strand and the column loop is unrolled by the period --
            // no letter fetch, no profile selection and no register shuffling between columns, and the template
            // ends fall on the loop's own boundary.
            int col = 0;
            for (; col < alen; ++col) {
                sweep_letter<R>(letter_from(tw, col), J, H, E, T, col, row0, geK, c0, row0g, rowbuf);
                ++cnt_cols;
            }
            int P0[R], P1[R], P2[R];
            pick_profile<R>(J, letter_from(tw, alen), geK, P0);
            pick_profile<R>(J, letter_from(tw, alen + 1), geK, P1);
            pick_profile<R>(J, letter_from(tw, alen + 2), geK, P2);
            for (;;) {
                sweep_at<R>(J, P0, H, E, T, col, row0, geK, c0, row0g, rowbuf);
                sweep_at<R>(J, P1, H, E, T, col + 1, row0, geK, c0, row0g, rowbuf);
                sweep_at<R>(J, P2, H, E, T, col + 2, row0, geK, c0, row0g, rowbuf);
                col += 3;
                cnt_cols += 3;
                const uint32_t ends_before = cnt_ends;
                if (template_end(col - 1) || col >= ncols || steady_exit(ends_before)) break;
            }
        } else {
            for (int col = 0; col < ncols; ++col) {
                sweep_letter<R>(letter_from(tw, col), J, H, E, T, col, row0, geK, c0, row0g, rowbuf);
                ++cnt_cols;
                if (col != next_end) continue;
                const uint32_t ends_before = cnt_ends;
                if (template_end(col) || (max_units > 0 && steady_exit(ends_before))) break;
            }
        }
    }
    const int lane_e = lane_now();
    if (a.stats != nullptr && lane_e < 7) {
        // one atomic per counter and wave (lanes 0..6 of one instruction), spread over SW_STAT_SLOTS lines;
        // the last one counts read-columns: columns swept x reads in the quad (empty slots of a partial quad excluded)
        const int n_trunk_cols = (int)(cnt_cols & 0xFFFFu), n_cont_cols = (int)(cnt_cols >> 16);
        const int vals[7] = {n_trunk_cols, n_cont_cols, (int)(cnt_ends & 0xFFu), (int)((cnt_ends >> 8) & 0xFFu),
                             (int)(cnt_ends >> 16), 1, (n_trunk_cols + n_cont_cols) * q_count};
        int v = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) v = lane_e == k ? vals[k] : v;
        atomicAdd(a.stats + (size_t)(blockIdx.x & (SW_STAT_SLOTS - 1)) * 8 + lane_e, (unsigned long long)v);
    }
    if ((lane_e >> 4) < q_count && (lane_e & 15) == 15) {
        const int64_t rd = (int64_t)a.perm[q_first + (lane_e >> 4)];
        int bestTag = best < 0 ? TREDGPU_TAG_NONE : best & 7;
        int bestU = 511 - ((best >> 3) & 511), bestS = best >> 12;
        if (too_long) bestTag = TREDGPU_TAG_INVALID, bestU = 0, bestS = 0;
        a.out_tag[rd] = (uint8_t)bestTag;
        a.out_h[rd] = (int16_t)(bestTag == TREDGPU_TAG_NONE ? 0 : bestU);
        a.out_score[rd] = (int16_t)(bestTag == TREDGPU_TAG_NONE ? 0 : bestS);
    }
}

// Strand classes and quad formation.  The exact 6-mer strand filter (see sw_cont_kernel) is evaluated per read
// here, once: bit s of the class = "strand s can reach the score filter for this read".  Reads that need no strand
// at all are finished on the spot (no candidate: tag NONE).  The others are packed four to a wavefront by
// (ladder, class, level) across the units of a batch -- a read's alignment does not depend on its wave-mates, so
// only the last quad of each bin can be partial (per-unit packing left 7 % of the read slots of the bench batch
// empty).  The level is the read's count of 6-mers present in the templates, in steps of 6 (32 levels): that count caps the
// read's score, a template of length T needs a score of min(L, T)/2, so reads of one level stop needing the
// trunk at about the same template and the wave can leave the strand together (strand exit in sw_cont_kernel).
// Bin b = (3 * ladder + k) * SW_LEVELS + level holds class {1, 3, 2}[k].
//   read_class_kernel   classes and levels, bin totals, each unit's offset inside its bins (one atomic per unit and bin)
//   bin_scan_kernel     bin -> first slot in the permutation / first quad; the quad count of the launch
//   scatter_kernel      each unit writes its reads' indices into its runs of the bins
//   fill_quads_kernel   one thread per quad: its bin by binary search over the quad offsets
constexpr int BIN_STRIDE = 16;   // ints: every bin counter on its own 64-byte line
constexpr int SW_LEVELS = 32;
constexpr int UNIT_BINS = 3 * SW_LEVELS;   // bins one unit can feed
__device__ __forceinline__ int class_slot(int cls) { return cls == 1 ? 0 : (cls == 3 ? 1 : 2); }

// The read's 6-mers present in the two strands' template bitmaps (windows with an N count as present).  `rec` = the
// read's packed record, bm = the two 128-word bitmaps (LDS).  One word of 16 bases per trip: the
// window ending at base j is one bit-field extract of the word (an alignbit with the previous word for j < 5), its
// two presence bits are collected into one mask per strand, the N rule and the first-five / past-the-end positions
// are applied to the masks once per word -- ~7 VALU instructions per base where the base-by-base loop (shift register,
// run length since the last N, two compares) took ~25: 0.35 -> 0.25 ms per 30 000 units (staging the records in LDS
// and 64-bit entries holding both strands' words were tried on top: no gain).
__device__ __forceinline__ void kmer_counts(const uint32_t* rec, int L, const uint32_t* bm, int& cnt0, int& cnt1) {
    const int nb = (L + 15) >> 4;
    uint32_t prev = 0, mprev = 0;
    cnt0 = cnt1 = 0;
    for (int wi = 0; wi < nb; ++wi) {
        const uint32_t w = rec[wi];
        const uint32_t m = (rec[nb + (wi >> 1)] >> ((wi & 1) * 16)) & 0xffffu;
        uint32_t b0 = 0, b1 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // bases j-5 .. j of this word, base j-5 in the low bits: the 6-mer's index is the low 12 bits of t (bitmap
            // word = bits 5 .. 11, bit = bits 0 .. 4 -- v_bfe_u32 takes its offset from the low five bits by itself)
            const uint32_t t = j > 5 ? w >> (2 * (j - 5)) : (j == 5 ? w : __builtin_amdgcn_alignbit(w, prev, 32 - 2 * (5 - j)));
            const uint32_t at = __builtin_amdgcn_ubfe(t, 5, 7);
            b0 = (__builtin_amdgcn_ubfe(bm[at], t, 1) << j) | b0;
            b1 = (__builtin_amdgcn_ubfe(bm[128 + at], t, 1) << j) | b1;
        }
        // N flags of bases -5 .. 15 of this word at bits 0 .. 20; window j holds an N when one of bits j .. j+5 is set
        const uint32_t mm = m << 5 | mprev >> 11;
        const uint32_t hn = mm | mm >> 1 | mm >> 2 | mm >> 3 | mm >> 4 | mm >> 5;
        // counted: read positions 5 .. L-1
        const int left = L - wi * 16;
        uint32_t cm = left >= 16 ? 0xffffu : (1u << left) - 1u;
        if (wi == 0) cm &= ~0x1fu;
        cnt0 += __builtin_popcount((b0 | hn) & cm);
        cnt1 += __builtin_popcount((b1 | hn) & cm);
        prev = w;
        mprev = m;
    }
}

__global__ __launch_bounds__(64) void read_class_kernel(SwArgs a, uint8_t* read_class, int32_t* unit_cnt, int32_t* bin_total) {
    const int g = blockIdx.x;
    if (g >= a.n_units) return;
    // (device-resident unit tables are not validated by the host: keep a bad index from leaving the tables)
    const int lad = min(max(a.unit_ladder[g], 0), a.n_ladders - 1);
    const LadderDesc* ld = a.ladders + lad;
    const bool full_dump = a.out_dump != nullptr;
    const int thr = min(a.p.mismatch, a.p.gap_open) >= 5 * a.p.match ? (30 + a.p.match - 1) / a.p.match - 5 : 0;
    const bool filt = !full_dump && thr > 0 && ld->kmer_ok != 0 && ld->max_units > 0;
    const int r0 = a.unit_read_off[g], r1 = a.unit_read_off[g + 1];
    // the two 4096-bit presence maps of the ladder, in LDS: every base of every read looks both up, and a
    // lookup in global memory (even an L1 hit) stalled the loop for its whole latency
    __shared__ uint32_t bm[2][128];
    __shared__ int hist[UNIT_BINS];
    if (filt) {
        for (int k = threadIdx.x; k < 256; k += (int)blockDim.x) bm[k >> 7][k & 127] = a.seqw[ld->kmer_off[k >> 7] + (k & 127)];
    }
    for (int k = threadIdx.x; k < UNIT_BINS; k += (int)blockDim.x) hist[k] = 0;
    __syncthreads();
    for (int rd = r0 + (int)threadIdx.x; rd < r1; rd += (int)blockDim.x) {
        int cls = ld->n_strands >= 2 ? 3 : 1;
        int level = SW_LEVELS - 1;
        const int L = a.read_len[rd];
        if (filt && L <= a.max_rows) {   // over-long reads go to the SW kernel, which flags them
            int cnt0, cnt1;
            kmer_counts(a.packed + a.read_off[rd], L, &bm[0][0], cnt0, cnt1);
            cls = (cnt0 >= thr ? 1 : 0) | (cnt1 >= thr ? 2 : 0);
            level = min(SW_LEVELS - 1, max(cls & 1 ? cnt0 : 0, cls & 2 ? cnt1 : 0) / 6);
        }
        read_class[rd] = (uint8_t)(cls | level << 2);
        if (cls == 0) {   // no strand can produce a candidate: bam_parser.py:171-172 "if not res: return"
            a.out_tag[rd] = TREDGPU_TAG_NONE;
            a.out_h[rd] = 0;
            a.out_score[rd] = 0;
        } else {
            atomicAdd(&hist[class_slot(cls) * SW_LEVELS + level], 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < UNIT_BINS; k += (int)blockDim.x) {
        // the unit's run inside each of its bins starts where the bin's total stood when the unit arrived
        const int n = hist[k];
        int at = 0;
        if (n) at = atomicAdd(bin_total + ((size_t)lad * UNIT_BINS + k) * BIN_STRIDE, n);
        unit_cnt[(size_t)g * UNIT_BINS + k] = at;
    }
}

// bins[b * BIN_STRIDE + 0] total reads, +1 first permutation slot, +2 first quad.
// One block: every thread sums a contiguous chunk of bins, the chunk totals are scanned in LDS.
__global__ __launch_bounds__(1024) void bin_scan_kernel(int32_t* bins, int n_bins, int32_t* n_quads) {
    __shared__ int cs[1024], cq[1024];
    const int t = threadIdx.x;
    const int chunk = (n_bins + 1023) / 1024;
    const int b0 = min(t * chunk, n_bins), b1 = min(b0 + chunk, n_bins);
    int slots = 0, quads = 0;
    for (int b = b0; b < b1; ++b) {
        const int n = bins[(size_t)b * BIN_STRIDE];
        slots += n;
        quads += (n + 3) >> 2;
    }
    cs[t] = slots; cq[t] = quads;
    __syncthreads();
    if (t == 0) {
        int s = 0, q = 0;
        for (int k = 0; k < 1024; ++k) { const int a_ = cs[k], b_ = cq[k]; cs[k] = s; cq[k] = q; s += a_; q += b_; }
        bins[(size_t)n_bins * BIN_STRIDE + 2] = q;   // sentinel for the binary search
        *n_quads = q;
    }
    __syncthreads();
    int slot = cs[t], quad = cq[t];
    for (int b = b0; b < b1; ++b) {
        int32_t* e = bins + (size_t)b * BIN_STRIDE;
        e[1] = slot;
        e[2] = quad;
        slot += e[0];
        quad += (e[0] + 3) >> 2;
    }
}

__global__ __launch_bounds__(64) void scatter_kernel(SwArgs a, const uint8_t* read_class, const int32_t* unit_cnt, int32_t* bins,
                                                       int32_t* perm) {
    const int g = blockIdx.x;
    if (g >= a.n_units) return;
    const int lad = min(max(a.unit_ladder[g], 0), a.n_ladders - 1);
    __shared__ int pos[UNIT_BINS];
    for (int k = threadIdx.x; k < UNIT_BINS; k += (int)blockDim.x)
        pos[k] = bins[((size_t)lad * UNIT_BINS + k) * BIN_STRIDE + 1] + unit_cnt[(size_t)g * UNIT_BINS + k];
    __syncthreads();
    const int r0 = a.unit_read_off[g], r1 = a.unit_read_off[g + 1];
    for (int r = r0 + (int)threadIdx.x; r < r1; r += (int)blockDim.x) {
        const int c = read_class[r] & 3, level = read_class[r] >> 2;
        if (c != 0) perm[atomicAdd(&pos[class_slot(c) * SW_LEVELS + level], 1)] = r;
    }
}

__global__ void fill_quads_kernel(const int32_t* bins, int n_bins, const int32_t* n_quads, Quad* quads) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= *n_quads) return;
    int lo = 0, hi = n_bins;   // largest b with first_quad[b] <= q (empty bins share their successor's offset)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bins[(size_t)mid * BIN_STRIDE + 2] <= q) lo = mid; else hi = mid;
    }
    const int32_t* e = bins + (size_t)lo * BIN_STRIDE;
    const int j = q - e[2];
    const int cls[3] = {1, 3, 2};
    Quad qd;
    qd.ladder = lo / UNIT_BINS;
    qd.first = e[1] + 4 * j;
    qd.count = min(4, e[0] - 4 * j);
    qd.strands = cls[(lo / SW_LEVELS) % 3];
    quads[q] = qd;
}

// bam_parser.py:256-287: histograms per unit; optional removal of REPT/REPT mate pairs (--norepeatpairs).
// remove_pairs_of_rept (:270-287) drops EVERY tagged read whose pair id carries two or more REPT reads, whatever
// its own tag (a supplementary FULL record of the same name goes too).  One workgroup per unit: REPT reads count
// their pair id in an open-addressing table in LDS, then every read looks its id up.  Units with more reads than
// the table holds comfortably fall back to comparing pairs of reads directly.
constexpr int PAIR_SLOTS = 4096;   // LDS table: 32 KiB (key + count)

__global__ __launch_bounds__(256) void mark_rept_pairs_kernel(const uint8_t* tag, const int32_t* read_pair_id,
                                                              const int32_t* unit_read_off, int32_t n_units,
                                                              uint8_t* drop) {
    __shared__ int keys[PAIR_SLOTS];
    __shared__ int cnts[PAIR_SLOTS];
    const int g = blockIdx.x;
    if (g >= n_units) return;
    const int r0 = unit_read_off[g], r1 = unit_read_off[g + 1];
    if (r1 - r0 > PAIR_SLOTS / 2) {
        for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
            const int pid = read_pair_id[i];
            int n = 0;
            // (no per-lane early exit at n == 2: hipcc 7.2 takes the exit test of the LAST loop trip for every lane
            //  of a loop whose lanes leave at different trips; the wave-uniform bound is also the simpler code)
            for (int j = r0; j < r1; ++j) n += (pid >= 0) & (read_pair_id[j] == pid) & (tag[j] == TREDGPU_TAG_REPT);
            drop[i] = n >= 2;
        }
        return;
    }
    for (int k = threadIdx.x; k < PAIR_SLOTS; k += blockDim.x) { keys[k] = -1; cnts[k] = 0; }
    __syncthreads();
    auto slot_of = [&](int pid, bool insert) {
        unsigned h = ((unsigned)pid * 2654435761u) >> 20;          // 12 bits
        for (;;) {
            const int k = (int)(h & (PAIR_SLOTS - 1));
            int cur = keys[k];
            if (cur == pid) return k;
            if (cur == -1) {
                if (!insert) return -1;
                cur = atomicCAS(&keys[k], -1, pid);
                if (cur == -1 || cur == pid) return k;
            }
            ++h;
        }
    };
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const int pid = read_pair_id[i];
        if (pid >= 0 && tag[i] == TREDGPU_TAG_REPT) atomicAdd(&cnts[slot_of(pid, true)], 1);
    }
    __syncthreads();
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const int pid = read_pair_id[i];
        const int k = pid >= 0 ? slot_of(pid, false) : -1;
        drop[i] = k >= 0 && cnts[k] >= 2;
    }
}

__global__ void tally_kernel(const uint8_t* tag, const int16_t* h, const int32_t* unit_read_off,
                             int32_t n_units, const uint8_t* drop, int32_t hist_stride,
                             int32_t* full_cnt, int32_t* pref_cnt, int32_t* rept_cnt) {
    const int g = blockIdx.x;
    if (g >= n_units) return;
    const int r0 = unit_read_off[g], r1 = unit_read_off[g + 1];
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const int t = tag[i];
        const int hh = h[i];
        if (t == TREDGPU_TAG_NONE || t == TREDGPU_TAG_HANG) continue;
        if (drop != nullptr && drop[i]) continue;
        if (hh < 0 || hh >= hist_stride) continue;
        int32_t* dst = t == TREDGPU_TAG_FULL ? full_cnt : (t == TREDGPU_TAG_REPT ? rept_cnt : pref_cnt);
        atomicAdd(dst + (int64_t)g * hist_stride + hh, 1);
    }
}

}  // namespace

hipError_t launch_build_quads(const SwArgs& a, uint8_t* read_class, int32_t* perm, Quad* quads, int32_t* n_quads,
                              int32_t* unit_cnt, int32_t* bins, int n_ladders, int64_t max_quads, hipStream_t s) {
    hipError_t e = hipMemsetAsync(n_quads, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    if (a.n_units <= 0) return hipSuccess;
    const int n_bins = UNIT_BINS * n_ladders;
    if ((e = hipMemsetAsync(bins, 0, sw_bin_bytes(n_ladders), s)) != hipSuccess) return e;
    read_class_kernel<<<a.n_units, 64, 0, s>>>(a, read_class, unit_cnt, bins);
    bin_scan_kernel<<<1, 1024, 0, s>>>(bins, n_bins, n_quads);
    scatter_kernel<<<a.n_units, 64, 0, s>>>(a, read_class, unit_cnt, bins, perm);
    fill_quads_kernel<<<(unsigned)((max_quads + 255) / 256), 256, 0, s>>>(bins, n_bins, n_quads, quads);
    return hipGetLastError();
}

size_t sw_unit_cnt_bytes(int n_units) { return (size_t)n_units * UNIT_BINS * sizeof(int32_t); }
int64_t sw_max_quads(int64_t n_reads, int n_ladders) { return n_reads / 4 + (int64_t)UNIT_BINS * n_ladders + 1; }
size_t sw_bin_bytes(int n_ladders) { return ((size_t)UNIT_BINS * n_ladders + 1) * BIN_STRIDE * sizeof(int32_t); }

hipError_t launch_sw_ladder(const SwArgs& a, int rows_per_lane, bool generic, int64_t max_quads, hipStream_t s) {
    if (max_quads <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)max_quads;   // one quad = one wavefront = one workgroup
    generic = generic || a.out_dump != nullptr;
    switch (rows_per_lane * 2 + (generic ? 1 : 0)) {
        // second parameter = waves per SIMD the register allocation is held to
        case 8: sw_cont_kernel<4, SW_WAVES_R4, false><<<blocks, 64, 0, s>>>(a); break;
        case 9: sw_cont_kernel<4, SW_WAVES_R4, true><<<blocks, 64, 0, s>>>(a); break;
        case 14: sw_cont_kernel<7, SW_WAVES_R7, false><<<blocks, 64, 0, s>>>(a); break;
        case 15: sw_cont_kernel<7, SW_WAVES_R7, true><<<blocks, 64, 0, s>>>(a); break;
        case 20: sw_cont_kernel<10, SW_WAVES_R10, false><<<blocks, 64, 0, s>>>(a); break;
        case 21: sw_cont_kernel<10, SW_WAVES_R10, true><<<blocks, 64, 0, s>>>(a); break;
        case 32: sw_cont_kernel<16, SW_WAVES_R16, false><<<blocks, 64, 0, s>>>(a); break;
        case 33: sw_cont_kernel<16, SW_WAVES_R16, true><<<blocks, 64, 0, s>>>(a); break;
        case 40: sw_cont_kernel<20, SW_WAVES_R20, false><<<blocks, 64, 0, s>>>(a); break;   // reads up to 320 bp (2 x 300 bp runs)
        case 41: sw_cont_kernel<20, SW_WAVES_R20, true><<<blocks, 64, 0, s>>>(a); break;
        case 64: sw_cont_kernel<32, SW_WAVES_R32, false><<<blocks, 64, 0, s>>>(a); break;   // reads up to 480 bp (512 rows: the packed values' nine row bits)
        case 65: sw_cont_kernel<32, SW_WAVES_R32, true><<<blocks, 64, 0, s>>>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_tally(const uint8_t* tag, const int16_t* h, int64_t n_reads,
                        const int32_t* unit_read_off, int32_t n_units, const int32_t* read_pair_id,
                        int32_t hist_stride, int32_t* full_cnt, int32_t* pref_cnt,
                        int32_t* rept_cnt, uint8_t* scratch_drop, hipStream_t s) {
    (void)n_reads;
    const size_t bytes = (size_t)n_units * hist_stride * sizeof(int32_t);
    hipError_t e;
    if ((e = hipMemsetAsync(full_cnt, 0, bytes, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(pref_cnt, 0, bytes, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(rept_cnt, 0, bytes, s)) != hipSuccess) return e;
    if (n_units <= 0) return hipSuccess;
    const uint8_t* drop = nullptr;
    if (read_pair_id != nullptr) {
        mark_rept_pairs_kernel<<<n_units, 256, 0, s>>>(tag, read_pair_id, unit_read_off, n_units, scratch_drop);
        drop = scratch_drop;
    }
    tally_kernel<<<n_units, 64, 0, s>>>(tag, h, unit_read_off, n_units, drop, hist_stride, full_cnt,
                                        pref_cnt, rept_cnt);
    return hipGetLastError();
}

}  // namespace tredgpu
