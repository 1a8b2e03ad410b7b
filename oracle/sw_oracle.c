/*
 * sw_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * Plain scalar C restatement of the reference's per-read template Smith-Waterman
 * classification, used as the parity checker for the HIP kernels:
 *
 *   oracle_sw_align      <- ssw_align()            /root/reference/src/ssw.c:780-871
 *                           forward pass           ssw.c:123-345 (byte) / :371-547 (word)
 *                           reverse pass           ssw.c:839-851
 *   oracle_base_code     <- Aligner.base_to_int    /root/reference/src/ssw_wrap.py:61,229-244
 *   score matrix         <- Aligner.set_mat        ssw_wrap.py:154-167
 *   oracle_build_ladder  <- BamParser._buildDB     /root/reference/tredparse/bam_parser.py:84-100, rc :448-450
 *   oracle_classify_read <- BamParser._parseReadSW bam_parser.py:123-182, get_hangs :102-121,
 *                           Aligner.align filter   ssw_wrap.py:214-220
 *   oracle_tally         <- tally_counts / rept    bam_parser.py:256-268
 *
 * The reference computes the DP with striped SSE2 vectors (Farrar) and saturating
 * unsigned bytes; this file computes the same recurrence cell by cell with plain ints.
 * Parity pin: tests/test_oracle_sw.py checks this file field-by-field against
 *   (a) committed golden vectors produced by the compiled reference (tests/golden/sw_pairs.npz),
 *   (b) oracle/_ref/libssw.so (the reference's own ssw.c compiled in place) when present.
 *
 * Result conventions follow s_align (ssw.h:42-52): all coordinates 0-based inclusive.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_TAG_NONE 0
#define ORACLE_TAG_FULL 1
#define ORACLE_TAG_PREF 2
#define ORACLE_TAG_POST 3
#define ORACLE_TAG_REPT 4
#define ORACLE_TAG_HANG 5

#define FLANKMATCH 9 /* bam_parser.py:30 */

/* ssw_wrap.py:61 -- A C G T -> 0..3 (either case), everything else -> 4 */
int oracle_base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

void oracle_encode(const char* s, int n, int8_t* out) {
    for (int i = 0; i < n; i++) out[i] = (int8_t)oracle_base_code(s[i]);
}

/* ssw_wrap.py:162-167: +match on the diagonal, -mismatch off it, 0 for any N row/col */
static inline int pair_score(int a, int b, int match, int mismatch) {
    if (a == 4 || b == 4) return 0;
    return a == b ? match : -mismatch;
}

/*
 * One DP sweep over ref columns c0, c0+step, ... (count n_cols), rows 0..L-1 of `read`.
 * Returns: best score, the FIRST column (in sweep order) whose column max strictly raised
 * the running max to its final value (ssw.c:281-288 / :491-495), and the smallest row of
 * that column holding the max (ssw.c:300-308 / :504-512).  If terminate >= 0 the sweep
 * stops after the first column whose column max equals it (ssw.c:294 / :500).
 * Zero-score conventions (ssw.c:143-145, 300-308): ref = -1, read = 0.
 */
static void dp_sweep(const int8_t* read, int L, const int8_t* ref, int c0, int step, int n_cols,
                     int match, int mismatch, int go, int ge, int terminate,
                     int* out_score, int* out_ref, int* out_read) {
    int* H = (int*)calloc((size_t)L + 1, sizeof(int));
    int* E = (int*)calloc((size_t)L + 1, sizeof(int));
    int* Hbest = (int*)calloc((size_t)L + 1, sizeof(int));
    int best = 0, best_ref = -1;
    int c = c0;
    for (int k = 0; k < n_cols; k++, c += step) {
        int rb = ref[c];
        int diag = 0; /* H[i-1][previous column] */
        int F = 0;
        int colmax = 0;
        for (int i = 0; i < L; i++) {
            int h = diag + pair_score(read[i], rb, match, mismatch);
            if (h < E[i]) h = E[i];
            if (h < F) h = F;
            if (h < 0) h = 0;
            diag = H[i];
            H[i] = h;
            if (h > colmax) colmax = h;
            int open = h - go;
            if (open < 0) open = 0;
            int e = E[i] - ge;
            if (e < 0) e = 0;
            E[i] = e > open ? e : open; /* E for the next column */
            int f = F - ge;
            if (f < 0) f = 0;
            F = f > open ? f : open; /* F for the next row */
        }
        if (colmax > best) {
            best = colmax;
            best_ref = c;
            memcpy(Hbest, H, (size_t)L * sizeof(int));
        }
        if (terminate >= 0 && colmax == terminate) break;
    }
    int best_read = 0;
    if (best > 0) {
        for (int i = 0; i < L; i++)
            if (Hbest[i] == best) { best_read = i; break; }
    }
    *out_score = best;
    *out_ref = best_ref;
    *out_read = best_read;
    free(H); free(E); free(Hbest);
}

/*
 * out = {score, ref_begin, ref_end, read_begin, read_end}   (ssw.c:824-851)
 * read/ref are base codes 0..4.
 */
int oracle_sw_align(const int8_t* read, int L, const int8_t* ref, int T,
                    int match, int mismatch, int go, int ge, int32_t out[5]) {
    int score, ref_end, read_end;
    dp_sweep(read, L, ref, 0, 1, T, match, mismatch, go, ge, -1, &score, &ref_end, &read_end);
    /* reverse pass: reversed read[0..read_end] against ref[ref_end..0], stop at first column
       whose max equals the forward score (ssw.c:839-846) */
    int n = read_end + 1;
    int8_t* rev = (int8_t*)malloc((size_t)n);
    for (int i = 0; i < n; i++) rev[i] = read[read_end - i];
    int s2, ref_begin, rrow;
    dp_sweep(rev, n, ref, ref_end, -1, ref_end + 1, match, mismatch, go, ge, score,
             &s2, &ref_begin, &rrow);
    free(rev);
    out[0] = score;
    out[1] = ref_begin;
    out[2] = ref_end;
    out[3] = read_end - rrow;
    out[4] = read_end;
    return 0;
}

/* batch of explicit (read, template) pairs; offsets are CSR into the code arrays */
int oracle_sw_pairs(const int8_t* reads, const int64_t* read_off, const int8_t* refs,
                    const int64_t* ref_off, const int32_t* pair_read, const int32_t* pair_ref,
                    int64_t n_pairs, int match, int mismatch, int go, int ge, int32_t* out,
                    int n_threads) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t p = 0; p < n_pairs; p++) {
        int r = pair_read[p], t = pair_ref[p];
        oracle_sw_align(reads + read_off[r], (int)(read_off[r + 1] - read_off[r]),
                        refs + ref_off[t], (int)(ref_off[t + 1] - ref_off[t]),
                        match, mismatch, go, ge, out + 5 * p);
    }
    return 0;
}

/* bam_parser.py:448-450 on base codes: complement A<->T, C<->G, N stays, then reverse */
static void revcomp_codes(const int8_t* s, int n, int8_t* out) {
    for (int i = 0; i < n; i++) {
        int c = s[n - 1 - i];
        out[i] = (int8_t)(c == 4 ? 4 : 3 - c);
    }
}

/*
 * bam_parser.py:84-100: for units = 1..max_units: target = prefix + repeat*units + suffix,
 * then rc(target); in that order.  Writes 2*max_units templates back to back into `out`
 * (capacity checked by the caller via oracle_ladder_size) and their offsets (2*max_units+1).
 */
int64_t oracle_ladder_size(int plen, int period, int slen, int max_units) {
    int64_t tot = 0;
    for (int u = 1; u <= max_units; u++) tot += 2 * (int64_t)(plen + slen + period * u);
    return tot;
}

int oracle_build_ladder(const int8_t* prefix, int plen, const int8_t* repeat, int period,
                        const int8_t* suffix, int slen, int max_units, int8_t* out,
                        int64_t* off) {
    int64_t pos = 0;
    int k = 0;
    for (int u = 1; u <= max_units; u++) {
        int T = plen + slen + period * u;
        int8_t* t = out + pos;
        memcpy(t, prefix, (size_t)plen);
        for (int j = 0; j < u; j++) memcpy(t + plen + j * period, repeat, (size_t)period);
        memcpy(t + plen + u * period, suffix, (size_t)slen);
        off[k++] = pos;
        pos += T;
        revcomp_codes(t, T, out + pos);
        off[k++] = pos;
        pos += T;
    }
    off[k] = pos;
    return 0;
}

/* bam_parser.py:102-121 */
static int get_hangs(const int32_t al[5], int T, int L) {
    int aL = al[1], aR = T - al[2] - 1;
    int bL = al[3], bR = L - al[4] - 1;
    int s1 = aR + bL, s2 = aL + bR, s3 = aL + aR, s4 = bL + bR;
    int m = s1 < s2 ? s1 : s2;
    if (s3 < m) m = s3;
    if (s4 < m) m = s4;
    return m;
}

/*
 * bam_parser.py:123-182 for one read against the ladder of one locus.
 *   max_units_global = ceil(READLEN / period)                      (bam_parser.py:73)
 *   clip != 0 -> REPT cut-off uses ceil(len(read)/period) instead  (bam_parser.py:154-155)
 * out = {tag, h, score}; tag ORACLE_TAG_NONE means "no candidate" (return at :171-172).
 * If per_template != NULL it receives, for each of the 2*max_units templates,
 * {score, ref_begin, ref_end, read_begin, read_end, tag_or_0} (6 ints) for debugging.
 */
int oracle_classify_read(const int8_t* read, int L, const int8_t* ladder, const int64_t* off,
                         int max_units_global, int period, int clip,
                         int match, int mismatch, int go, int ge,
                         int32_t out[3], int32_t* per_template) {
    int best_score = -1, best_units = 0, best_tag = ORACLE_TAG_NONE;
    int n_templates = 2 * max_units_global;
    for (int k = 0; k < n_templates; k++) {
        int units = k / 2 + 1;
        const int8_t* target = ladder + off[k];
        int T = (int)(off[k + 1] - off[k]);
        int32_t al[5];
        oracle_sw_align(read, L, target, T, match, mismatch, go, ge, al);
        int tag = ORACLE_TAG_NONE;
        int min_len = (L < T ? L : T) / 2;               /* bam_parser.py:133 (py2 int division) */
        int min_score = min_len > 30 ? min_len : 30;     /* :134 */
        int match_len = al[4] - al[3] + 1;               /* ssw_wrap.py:215 */
        if (al[0] >= min_score && match_len >= min_len) { /* ssw_wrap.py:217 */
            int prefix_read = al[1] < FLANKMATCH;                /* :139 */
            int suffix_read = al[2] > T - FLANKMATCH - 1;        /* :140 */
            int hang_read = get_hangs(al, T, L) >= FLANKMATCH;   /* :141-142 */
            int max_units = clip ? (L + period - 1) / period : max_units_global; /* :154-155 */
            if (hang_read) tag = ORACLE_TAG_HANG;
            else if (prefix_read) tag = suffix_read ? ORACLE_TAG_FULL : ORACLE_TAG_PREF;
            else if (suffix_read) tag = ORACLE_TAG_POST;
            else if (units >= max_units - 1 && units * period <= L) tag = ORACLE_TAG_REPT;
            if (tag != ORACLE_TAG_NONE) {
                /* max(res, key=(score, -units)); python max keeps the FIRST maximal element */
                if (al[0] > best_score || (al[0] == best_score && units < best_units)) {
                    best_score = al[0];
                    best_units = units;
                    best_tag = tag;
                }
            }
        }
        if (per_template) {
            memcpy(per_template + 6 * k, al, 5 * sizeof(int32_t));
            per_template[6 * k + 5] = tag;
        }
    }
    out[0] = best_tag;
    out[1] = best_tag == ORACLE_TAG_NONE ? 0 : best_units;
    out[2] = best_tag == ORACLE_TAG_NONE ? 0 : best_score;
    return 0;
}

/*
 * Batch: reads (CSR codes) each belonging to a group; each group names a locus ladder.
 * ladders are stored back to back: ladder_tmpl_off[locus] indexes into tmpl_off (CSR of
 * templates), i.e. templates of locus g are tmpl_off[ladder_tmpl_off[g] .. ladder_tmpl_off[g+1]].
 */
int oracle_classify_batch(const int8_t* reads, const int64_t* read_off, const int32_t* read_locus,
                          const int32_t* read_maxunits, int64_t n_reads,
                          const int8_t* tmpl_codes, const int64_t* tmpl_off,
                          const int64_t* ladder_tmpl_off, const int32_t* locus_period,
                          int clip, int match, int mismatch, int go, int ge,
                          int32_t* out, int n_threads) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t r = 0; r < n_reads; r++) {
        int g = read_locus[r];
        const int64_t* off = tmpl_off + ladder_tmpl_off[g];
        oracle_classify_read(reads + read_off[r], (int)(read_off[r + 1] - read_off[r]),
                             tmpl_codes, off, read_maxunits[r], locus_period[g], clip,
                             match, mismatch, go, ge, out + 3 * r, NULL);
    }
    return 0;
}
