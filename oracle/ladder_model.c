/*
 * ladder_model.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU model of the *algorithm* the HIP kernel uses (tredparse_amd/csrc/sw_ladder.hip), so the
 * two exact shortcuts can be validated on the CPU against the plain restatement (sw_oracle.c) and
 * the compiled reference before any GPU time is spent:
 *
 *  1. shared-prefix ladder: all templates prefix+repeat*u+suffix (bam_parser.py:91-93) share the
 *     trunk prefix+repeat*max_units; a forward DP column depends only on columns to its left, so
 *     the trunk is swept once and the |suffix| branch columns are swept per u from a copy of the
 *     trunk state at column |prefix|+period*u-1.
 *  2. one-pass begin coordinates: every DP value is one int32 = score<<18 | start_col<<9 | start_row,
 *     and integer max then selects (score, then the largest start column, then the largest start
 *     row) -- which is what the reference's reverse pass (ssw.c:839-851) reports: the first column
 *     walking left from ref_end whose reversed-DP max equals the score, and the smallest reversed
 *     row in it.  End coordinates use a second key score<<18 | (511-col)<<9 | (511-row) whose max is
 *     the reference's "first column reaching the max, smallest row in it" (ssw.c:281-288,300-308).
 *
 * This model is not a restatement of the reference and is not the parity oracle; it exists to
 * check the kernel's algorithm.  Limits: template length <= 511, read length <= 511.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KSH 18
#define KONE (1 << KSH)
#define PAYMASK (KONE - 1)

static inline int imax(int a, int b) { return a > b ? a : b; }

static inline int pscore(int a, int b, int match, int mismatch) {
    if (a == 4 || b == 4) return 0;
    return a == b ? match : -mismatch;
}

typedef struct {
    int* H;   /* packed H of the last swept column */
    int* E;   /* packed E for the next column */
    int bestkey;
    int beststart;
} lstate;

/* sweep one column with reference letter rb at template column index col */
static void column(lstate* s, const int8_t* read, int L, int rb, int col,
                   int match, int mismatch, int go, int ge, int* ht) {
    int diag = 0;
    /* pass 1: H without the vertical-gap term */
    for (int i = 0; i < L; i++) {
        int fresh = (col << 9) | i;
        int d = imax(diag, fresh);
        int v = d + pscore(read[i], rb, match, mismatch) * KONE;
        v = imax(v, s->E[i]);
        diag = s->H[i];
        ht[i] = v;
    }
    /* pass 2: F as a running max (the kernel does this with a wave prefix scan) */
    int F = -(1 << 30);
    int revcol = (511 - col) << 9;
    for (int i = 0; i < L; i++) {
        int h = imax(ht[i], F);
        F = imax(F - ge * KONE, ht[i] - go * KONE);
        s->H[i] = h;
        s->E[i] = imax(s->E[i] - ge * KONE, h - go * KONE);
        int key = (h & ~PAYMASK) | revcol | (511 - i);
        if (key > s->bestkey) {
            s->bestkey = key;
            s->beststart = h & PAYMASK;
        }
    }
}

static void emit(const lstate* s, int32_t out[5]) {
    int score = s->bestkey >> KSH;
    if (score <= 0) { /* sentinel of sw_oracle.c for "no alignment" */
        out[0] = 0; out[1] = -1; out[2] = -1; out[3] = 0; out[4] = 0;
        return;
    }
    out[0] = score;
    out[1] = (s->beststart >> 9) & 511;
    out[2] = 511 - ((s->bestkey >> 9) & 511);
    out[3] = s->beststart & 511;
    out[4] = 511 - (s->bestkey & 511);
}

/*
 * One read against one strand's ladder (A + rep*u + B, u = 1..max_units).
 * out: max_units x 5 ints {score, ref_begin, ref_end, read_begin, read_end}.
 */
int ladder_model_strand(const int8_t* read, int L, const int8_t* A, int alen,
                        const int8_t* rep, int period, const int8_t* B, int blen,
                        int max_units, int match, int mismatch, int go, int ge, int32_t* out) {
    if (L > 511 || alen + blen + period * max_units > 511) return -1;
    lstate t, b;
    t.H = (int*)calloc((size_t)L, sizeof(int));
    t.E = (int*)calloc((size_t)L, sizeof(int));
    b.H = (int*)malloc((size_t)L * sizeof(int));
    b.E = (int*)malloc((size_t)L * sizeof(int));
    int* ht = (int*)malloc((size_t)L * sizeof(int));
    t.bestkey = 0;
    t.beststart = 0;
    int ncols = alen + period * max_units;
    for (int c = 0; c < ncols; c++) {
        int rb = c < alen ? A[c] : rep[(c - alen) % period];
        column(&t, read, L, rb, c, match, mismatch, go, ge, ht);
        if (c >= alen + period - 1 && (c - alen + 1) % period == 0) {
            int u = (c - alen + 1) / period;
            memcpy(b.H, t.H, (size_t)L * sizeof(int));
            memcpy(b.E, t.E, (size_t)L * sizeof(int));
            b.bestkey = t.bestkey;
            b.beststart = t.beststart;
            for (int k = 0; k < blen; k++)
                column(&b, read, L, B[k], c + 1 + k, match, mismatch, go, ge, ht);
            emit(&b, out + 5 * (u - 1));
        }
    }
    free(t.H); free(t.E); free(b.H); free(b.E); free(ht);
    return 0;
}
