/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY.  Links against oracle/_ref/libssw.so, i.e. the
 * reference's own src/ssw.c compiled in place (never copied into this repo), and drives it the
 * way the reference's Python wrapper does, but without the interpreter in the loop:
 *
 *   ref_sw_pairs      one ssw_init + ssw_align + 2 destroys per (read, template) pair with
 *                     score_size=2, flag=1, filters=0, filterd=0, maskLen=len/2 (>30) else 15
 *                     -- exactly Aligner.align, /root/reference/src/ssw_wrap.py:177-227
 *   ref_classify_batch  _parseReadSW (/root/reference/tredparse/bam_parser.py:123-182) with the
 *                     alignments coming from the reference library.
 *
 * Used to (1) generate/verify golden SW vectors, (2) time the reference's CPU path natively as
 * bench.py's cpu_baseline kind="reference".
 *
 * The reference's CIGAR pass can fault: banded_sw (/root/reference/src/ssw.c:549-633) doubles its band until the banded
 * score reaches the striped one (`width_d * readLen * 3` in 32-bit ints, :580); with cheap gaps and a periodic 300-base read against a
 * 336-base template it never gets there and runs off its buffers (found by tools/fuzz_parity.py, seed 20271201 round
 * 334: scoring 1/3/2/2, a (CTG)n read on the DM1 ladder).  The product computes no CIGAR.  So that a campaign
 * survives such an input, every ssw_align call runs under a SIGSEGV / SIGBUS guard: a faulting call is abandoned
 * (its allocations leak) and reported as REF_CRASHED -- per pair in ref_sw_pairs, as tag -1 for the read in
 * ref_classify_batch -- and the callers leave those out of the comparison and count them.  The fault is a heap
 * overrun, so nothing else the same process computed in that call is trusted either: oracle/pyoracle.py computes the
 * call's other items again in a fresh process.
 */
#include <setjmp.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ssw.h" /* from $(REF)/src via -I, not vendored */
#ifdef _OPENMP
#include <omp.h>
#endif

static void make_mat(int match, int mismatch, int8_t mat[25]) {
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            mat[a * 5 + b] = (int8_t)((a == 4 || b == 4) ? 0 : (a == b ? match : -mismatch));
}

#define REF_CRASHED (-9998)
static __thread sigjmp_buf guard_jmp;
static __thread volatile sig_atomic_t guard_armed;
static struct sigaction guard_old[2];

static void guard_handler(int sig) {
    if (guard_armed) siglongjmp(guard_jmp, 1);
    /* not ours: hand the signal back to whoever had it (Python's faulthandler, or the default action) */
    sigaction(sig, &guard_old[sig == SIGSEGV ? 0 : 1], NULL);
    raise(sig);
}

/* installed ONCE per process and left in place (two Python threads may be inside the driver at the same time: a handler
 * that is put in and taken out per call would save and restore the other call's handler); a fault that is not the
 * driver's goes back to whoever had the signal before */
#include <pthread.h>
static pthread_once_t guard_once = PTHREAD_ONCE_INIT;
static void guard_install_once(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = guard_handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGSEGV, &sa, &guard_old[0]);
    sigaction(SIGBUS, &sa, &guard_old[1]);
}
static void guard_install(void) { pthread_once(&guard_once, guard_install_once); }
static void guard_remove(void) {}

static void ref_align_one(const int8_t* read, int L, const int8_t* ref, int T, const int8_t* mat,
                          int go, int ge, int32_t out[5]) {
    s_profile* p = ssw_init(read, L, mat, 5, 2);
    int mask_len = L > 30 ? L / 2 : 15;
    s_align* volatile a = NULL;
    if (sigsetjmp(guard_jmp, 1) == 0) {
        guard_armed = 1;
        a = ssw_align(p, ref, T, (uint8_t)go, (uint8_t)ge, 1, 0, 0, mask_len);
        guard_armed = 0;
    } else {   /* the reference faulted inside ssw_align */
        guard_armed = 0;
        out[0] = out[1] = out[2] = out[3] = out[4] = REF_CRASHED;
        return;
    }
    if (a) {
        out[0] = a->score1;
        out[1] = a->ref_begin1;
        out[2] = a->ref_end1;
        out[3] = a->read_begin1;
        out[4] = a->read_end1;
        align_destroy(a);
    } else {
        out[0] = out[1] = out[2] = out[3] = out[4] = -9999;
    }
    init_destroy(p);
}

int ref_sw_pairs(const int8_t* reads, const int64_t* read_off, const int8_t* refs,
                 const int64_t* ref_off, const int32_t* pair_read, const int32_t* pair_ref,
                 int64_t n_pairs, int match, int mismatch, int go, int ge, int32_t* out,
                 int n_threads) {
    int8_t mat[25];
    make_mat(match, mismatch, mat);
    guard_install();
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t k = 0; k < n_pairs; k++) {
        int r = pair_read[k], t = pair_ref[k];
        ref_align_one(reads + read_off[r], (int)(read_off[r + 1] - read_off[r]),
                      refs + ref_off[t], (int)(ref_off[t + 1] - ref_off[t]), mat, go, ge,
                      out + 5 * k);
    }
    guard_remove();
    return 0;
}

#define FLANKMATCH 9
enum { TAG_NONE = 0, TAG_FULL, TAG_PREF, TAG_POST, TAG_REPT, TAG_HANG };

static int hangs(const int32_t al[5], int T, int L) {
    int aL = al[1], aR = T - al[2] - 1, bL = al[3], bR = L - al[4] - 1;
    int m = aR + bL;
    if (aL + bR < m) m = aL + bR;
    if (aL + aR < m) m = aL + aR;
    if (bL + bR < m) m = bL + bR;
    return m;
}

int ref_classify_batch(const int8_t* reads, const int64_t* read_off, const int32_t* read_locus,
                       const int32_t* read_maxunits, int64_t n_reads,
                       const int8_t* tmpl_codes, const int64_t* tmpl_off,
                       const int64_t* ladder_tmpl_off, const int32_t* locus_period,
                       int clip, int match, int mismatch, int go, int ge,
                       int32_t* out, int n_threads) {
    int8_t mat[25];
    make_mat(match, mismatch, mat);
    guard_install();
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t r = 0; r < n_reads; r++) {
        int g = read_locus[r];
        const int64_t* off = tmpl_off + ladder_tmpl_off[g];
        const int8_t* read = reads + read_off[r];
        int L = (int)(read_off[r + 1] - read_off[r]);
        int period = locus_period[g];
        int best_score = -1, best_units = 0, best_tag = TAG_NONE, crashed = 0;
        for (int k = 0; k < 2 * read_maxunits[r]; k++) {
            int units = k / 2 + 1;
            int T = (int)(off[k + 1] - off[k]);
            int32_t al[5];
            ref_align_one(read, L, tmpl_codes + off[k], T, mat, go, ge, al);
            if (al[0] == REF_CRASHED) { crashed = 1; break; }
            int min_len = (L < T ? L : T) / 2;
            int min_score = min_len > 30 ? min_len : 30;
            if (!(al[0] >= min_score && al[4] - al[3] + 1 >= min_len)) continue;
            int tag = TAG_NONE;
            int max_units = clip ? (L + period - 1) / period : read_maxunits[r];
            if (hangs(al, T, L) >= FLANKMATCH) tag = TAG_HANG;
            else if (al[1] < FLANKMATCH) tag = al[2] > T - FLANKMATCH - 1 ? TAG_FULL : TAG_PREF;
            else if (al[2] > T - FLANKMATCH - 1) tag = TAG_POST;
            else if (units >= max_units - 1 && units * period <= L) tag = TAG_REPT;
            if (tag == TAG_NONE) continue;
            if (al[0] > best_score || (al[0] == best_score && units < best_units)) {
                best_score = al[0]; best_units = units; best_tag = tag;
            }
        }
        out[3 * r] = crashed ? -1 : best_tag;
        out[3 * r + 1] = crashed || best_tag == TAG_NONE ? 0 : best_units;
        out[3 * r + 2] = crashed || best_tag == TAG_NONE ? 0 : best_score;
    }
    guard_remove();
    return 0;
}
