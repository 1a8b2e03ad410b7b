// sw_ladder.hip -- template-ladder Smith-Waterman + read tagging for gfx950 (MI355X, CDNA4).
//
// Replaces, for a whole batch of reads in one launch:
//   /root/reference/src/ssw.c:780-871        ssw_align (forward pass :123-345/:371-547, reverse pass :839-851)
//   /root/reference/src/ssw_wrap.py:177-227  Aligner.align incl. the score / length filter :214-220
//   /root/reference/tredparse/bam_parser.py:123-182  _parseReadSW (hangs :102-121, tags :139-168, arg-max :174)
//
// Design (not a translation of the striped SSE2 code):
//  * one wavefront = four reads of one sample x locus unit, 16 lanes per read, R consecutive read
//    rows per lane (R = ceil(maxlen/16)); the DP column lives in registers, lanes talk through DPP
//    row shifts only (no LDS, no barriers), so a DPP row (16 lanes) is exactly one alignment.
//  * the vertical-gap term F is an exclusive max-plus prefix scan over the 16 lanes (4 DPP steps)
//    instead of Farrar's data-dependent lazy-F loop.
//  * shared-prefix ladder: templates prefix+repeat*u+suffix (u=1..max_units) share the trunk
//    prefix+repeat*max_units; the trunk is swept once and the |suffix| branch columns are swept per u
//    from a register copy of the trunk state -- 5.3x fewer cells than 2*max_units independent
//    alignments for period 3 / 150 bp, and bit-identical because a forward column depends only on
//    the columns to its left.
//  * begin coordinates without the reference's reverse pass: every DP value is one int32
//    score<<18 | start_col<<9 | start_row; integer max then picks (score, largest start column,
//    largest start row), which is what the reverse pass reports (first column walking left whose
//    max equals the score, smallest reversed row).  End coordinates use the key
//    score<<18 | (511-col)<<9 | (511-row): its max is "first column reaching the max, smallest row".
//    Both rules are validated against the compiled reference in oracle/ladder_model.c's tests.
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

// inclusive max scan over the 16 lanes of a DPP row
__device__ __forceinline__ int row_scan_max(int x) {
    x = max(x, dpp_row_shr<0x111>(NEG, x));
    x = max(x, dpp_row_shr<0x112>(NEG, x));
    x = max(x, dpp_row_shr<0x114>(NEG, x));
    x = max(x, dpp_row_shr<0x118>(NEG, x));
    return x;
}

template <int CTRL>
__device__ __forceinline__ void pair_step(int& k, int& s) {
    int tk = dpp_row_shr<CTRL>(0, k);
    int ts = dpp_row_shr<CTRL>(0, s);
    bool c = tk > k;
    k = c ? tk : k;
    s = c ? ts : s;
}

template <int R>
struct Rows {
    int bc[R];  // base code compared with the template letter: 0..3, 4 = N, 5 = padding row
    int xk[R];  // addend when the letter differs: -mismatch*K, 0 for N rows, PADNEG for padding
    int zk[R];  // addend in an N template column: 0, PADNEG for padding
    int rr[R];  // 511 - row index
};

// One DP column for all rows of the four alignments in this wave.
//   H: packed H of the previous column (in) / this column (out);  E: packed E for this column (in) /
//   the next column (out).  letter is wave-uniform.
template <int R>
__device__ __forceinline__ void sweep_column(const Rows<R>& J, int (&H)[R], int (&E)[R],
                                             int& bestkey, int& beststart, int letter, int col,
                                             int row0, int mK, int goK, int geK, int shl,
                                             int geRK) {
    // diagonal input of this lane's first row: last row of the lane above, previous column
    const int hup = dpp_row_shr<0x111>(0, H[R - 1]);
    const int colbits = col << 9;
    const bool ncol = letter == 4;
    int ht[R], u[R];
    int diag = hup;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = ncol ? J.zk[r] : (J.bc[r] == letter ? mK : J.xk[r]);
        const int t1 = diag + s;                    // extend the alignment ending at (row-1, col-1)
        const int t2 = (row0 + r) + colbits + s;    // or start a new one here
        const int v = max(max(t1, t2), E[r]);
        diag = H[r];
        ht[r] = v;
        u[r] = v - goK;
    }
    // F entering the row below this lane, from this lane's rows only
    int a = u[0];
#pragma unroll
    for (int r = 1; r < R; ++r) a = max(a - geK, u[r]);
    // exclusive max-plus scan across the 16 lanes: F entering this lane's first row
    const int p = row_scan_max(a + shl);
    int f = dpp_row_shr<0x111>(NEG, p) - shl + geRK;
    const int revcol = (511 - col) << 9;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int h = max(ht[r], f);
        f = max(f - geK, u[r]);
        H[r] = h;
        E[r] = max(E[r] - geK, h - goK);
        const int cand = (h & ~PAYMASK) | revcol | J.rr[r];
        const bool c = cand > bestkey;
        bestkey = c ? cand : bestkey;
        beststart = c ? h : beststart;
    }
}

template <int R>
__global__ __launch_bounds__(256) void sw_ladder_kernel(SwArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const int nq = *a.n_quads;
    if (q >= nq) return;
    const int q_unit = __builtin_amdgcn_readfirstlane(a.quads[q].unit);
    const int q_read0 = __builtin_amdgcn_readfirstlane(a.quads[q].read0);
    const int q_count = __builtin_amdgcn_readfirstlane(a.quads[q].count);
    const LadderDesc* ld = a.ladders + __builtin_amdgcn_readfirstlane(a.unit_ladder[q_unit]);
    const int period = __builtin_amdgcn_readfirstlane(ld->period);
    const int max_units = __builtin_amdgcn_readfirstlane(ld->max_units);
    const int n_strands = __builtin_amdgcn_readfirstlane(ld->n_strands);

    const int job = lane >> 4, jl = lane & 15;
    const bool valid = job < q_count;
    const int64_t rd = (int64_t)q_read0 + job;
    int L = valid ? a.read_len[rd] : 0;
    const bool too_long = L > 16 * R;  // not representable in this instantiation: flagged, not aligned
    if (too_long) L = 0;
    const int64_t off = valid ? a.read_off[rd] : 0;
    const int row0 = jl * R;

    const int mK = a.p.match * KONE;
    const int xK = -a.p.mismatch * KONE;
    const int goK = a.p.gap_open * KONE;
    const int geK = a.p.gap_extend * KONE;
    const int geRK = geK * R;
    const int shl = geRK * jl;
    const int flank = a.p.flank;

    Rows<R> J;
    {
        const int nb = (L + 15) >> 4;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r;
            J.rr[r] = 511 - i;
            if (i < L) {
                const uint32_t w = a.packed[off + (i >> 4)];
                const uint32_t m = a.packed[off + nb + (i >> 5)];
                const bool isn = (m >> (i & 31)) & 1u;
                J.bc[r] = isn ? 4 : (int)((w >> ((i & 15) * 2)) & 3u);
                J.xk[r] = isn ? 0 : xK;
                J.zk[r] = 0;
            } else {
                J.bc[r] = 5;
                J.xk[r] = PADNEG;
                J.zk[r] = PADNEG;
            }
        }
    }
    // REPT cut-off: per-read ceil(L/period) with --useclippedreads, else the ladder's (bam_parser.py:154-155)
    const int mu_rept = a.p.clip ? (L + period - 1) / period : max_units;

    int bestS = -1, bestU = 0, bestTag = TREDGPU_TAG_NONE;
    int16_t* dump = nullptr;
    if (a.out_dump != nullptr && valid) dump = a.out_dump + rd * (int64_t)a.dump_templates * 6;

    for (int s = 0; s < n_strands; ++s) {
        const int8_t* trunk = a.seq + __builtin_amdgcn_readfirstlane(ld->trunk_off[s]);
        const int8_t* branch = a.seq + __builtin_amdgcn_readfirstlane(ld->branch_off[s]);
        const int alen = __builtin_amdgcn_readfirstlane(ld->alen[s]);
        const int blen = __builtin_amdgcn_readfirstlane(ld->blen[s]);
        const int ncols = alen + period * max_units;
        int H[R], E[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { H[r] = 0; E[r] = 0; }
        int bestkey = 0, beststart = 0;
        int next_branch = max_units > 0 ? alen + period - 1 : alen - 1;
        int u = max_units > 0 ? 1 : 0;
        for (int c = 0; c < ncols; ++c) {
            const int letter = trunk[c];
            sweep_column<R>(J, H, E, bestkey, beststart, letter, c, row0, mK, goK, geK, shl, geRK);
            if (c != next_branch) continue;
            // ---- template u ends here on the trunk: run its branch from a copy of the state ----
            int Hb[R], Eb[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { Hb[r] = H[r]; Eb[r] = E[r]; }
            int bk = bestkey, bs = beststart;
            for (int k = 0; k < blen; ++k)
                sweep_column<R>(J, Hb, Eb, bk, bs, branch[k], c + 1 + k, row0, mK, goK, geK, shl, geRK);
            // best cell of the alignment -> lane 15 of the read's DPP row
            pair_step<0x111>(bk, bs);
            pair_step<0x112>(bk, bs);
            pair_step<0x114>(bk, bs);
            pair_step<0x118>(bk, bs);
            const int score = bk >> KSH;
            const int ref_end = 511 - ((bk >> 9) & 511), read_end = 511 - (bk & 511);
            const int ref_begin = (bs >> 9) & 511, read_begin = bs & 511;
            const int T = alen + period * u + blen;
            const int min_len = min(L, T) >> 1;                 // bam_parser.py:133
            const int min_score = max(min_len, 30);             // :134
            const bool pass = score >= min_score && (read_end - read_begin + 1) >= min_len;  // ssw_wrap.py:217
            const int aL = ref_begin, aR = T - ref_end - 1, bL = read_begin, bR = L - read_end - 1;
            const int hang = min(min(aR + bL, aL + bR), min(aL + aR, bL + bR));  // bam_parser.py:113-121
            const bool prefix_read = ref_begin < flank;                           // :139
            const bool suffix_read = ref_end > T - flank - 1;                     // :140
            int tag;
            if (hang >= flank) tag = TREDGPU_TAG_HANG;
            else if (prefix_read) tag = suffix_read ? TREDGPU_TAG_FULL : TREDGPU_TAG_PREF;
            else if (suffix_read) tag = TREDGPU_TAG_POST;
            else if (u >= mu_rept - 1 && u * period <= L) tag = TREDGPU_TAG_REPT;
            else tag = TREDGPU_TAG_NONE;
            if (!pass) tag = TREDGPU_TAG_NONE;
            // max(res, key=(score, -units)), first maximal element in db order (bam_parser.py:174)
            const bool better = tag != TREDGPU_TAG_NONE && (score > bestS || (score == bestS && u < bestU));
            bestS = better ? score : bestS;
            bestU = better ? u : bestU;
            bestTag = better ? tag : bestTag;
            if (dump != nullptr && jl == 15) {
                const int k = max_units > 0 ? 2 * (u - 1) + s : 0;
                if (k < a.dump_templates) {
                    int16_t* d = dump + k * 6;
                    const bool hit = score > 0;
                    d[0] = (int16_t)(hit ? score : 0);
                    d[1] = (int16_t)(hit ? ref_begin : -1);
                    d[2] = (int16_t)(hit ? ref_end : -1);
                    d[3] = (int16_t)(hit ? read_begin : 0);
                    d[4] = (int16_t)(hit ? read_end : 0);
                    d[5] = (int16_t)tag;
                }
            }
            next_branch += period;
            ++u;
        }
    }
    if (valid && jl == 15) {
        if (too_long) bestTag = TREDGPU_TAG_INVALID, bestU = 0, bestS = 0;
        a.out_tag[rd] = (uint8_t)bestTag;
        a.out_h[rd] = (int16_t)(bestTag == TREDGPU_TAG_NONE ? 0 : bestU);
        a.out_score[rd] = (int16_t)(bestTag == TREDGPU_TAG_NONE ? 0 : bestS);
    }
}

__global__ void build_quads_kernel(const int32_t* unit_read_off, int32_t n_units, Quad* quads,
                                   int32_t* n_quads) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_units) return;
    const int r0 = unit_read_off[g], n = unit_read_off[g + 1] - r0;
    const int nq = (n + 3) >> 2;
    if (nq == 0) return;
    const int base = atomicAdd(n_quads, nq);
    for (int k = 0; k < nq; ++k) {
        Quad qd;
        qd.unit = g;
        qd.read0 = r0 + 4 * k;
        qd.count = min(4, n - 4 * k);
        qd.pad = 0;
        quads[base + k] = qd;
    }
}

// bam_parser.py:256-287: histograms per unit; optional removal of REPT/REPT mate pairs.
__global__ void mark_rept_pairs_kernel(const uint8_t* tag, const int32_t* read_pair_id,
                                       const int32_t* unit_read_off, int32_t n_units,
                                       uint8_t* drop) {
    const int g = blockIdx.x;
    if (g >= n_units) return;
    const int r0 = unit_read_off[g], r1 = unit_read_off[g + 1];
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        uint8_t d = 0;
        const int pid = read_pair_id[i];
        if (tag[i] == TREDGPU_TAG_REPT && pid >= 0) {
            for (int j = r0; j < r1; ++j)
                if (j != i && read_pair_id[j] == pid && tag[j] == TREDGPU_TAG_REPT) { d = 1; break; }
        }
        drop[i] = d;
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

hipError_t launch_build_quads(const int32_t* unit_read_off, int32_t n_units, Quad* quads,
                              int32_t* n_quads, hipStream_t s) {
    hipError_t e = hipMemsetAsync(n_quads, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    if (n_units <= 0) return hipSuccess;
    build_quads_kernel<<<(n_units + 255) / 256, 256, 0, s>>>(unit_read_off, n_units, quads, n_quads);
    return hipGetLastError();
}

hipError_t launch_sw_ladder(const SwArgs& a, int rows_per_lane, int64_t max_quads, hipStream_t s) {
    if (max_quads <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((max_quads + 3) / 4);
    switch (rows_per_lane) {
        case 4: sw_ladder_kernel<4><<<blocks, 256, 0, s>>>(a); break;
        case 7: sw_ladder_kernel<7><<<blocks, 256, 0, s>>>(a); break;
        case 10: sw_ladder_kernel<10><<<blocks, 256, 0, s>>>(a); break;
        case 16: sw_ladder_kernel<16><<<blocks, 256, 0, s>>>(a); break;
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
        mark_rept_pairs_kernel<<<n_units, 64, 0, s>>>(tag, read_pair_id, unit_read_off, n_units, scratch_drop);
        drop = scratch_drop;
    }
    tally_kernel<<<n_units, 64, 0, s>>>(tag, h, unit_read_off, n_units, drop, hist_stride, full_cnt,
                                        pref_cnt, rept_cnt);
    return hipGetLastError();
}

}  // namespace tredgpu
