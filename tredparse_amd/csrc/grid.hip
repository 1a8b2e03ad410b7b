// grid.hip -- (h1,h2) allele-pair likelihood grid + paired-end KDE for gfx950 (fp64).
//
// Replaces /root/reference/tredparse/models.py:
//   pdf_spanning :149-168, pdf_partial :170-180, get_alpha :182-190, evaluate_spanning :192-198,
//   evaluate_partial :200-207, evaluate_rept :209-221, evaluate :223-302, calc_CI :319-340,
//   calc_PP :342-368, safe_log :418-423, PEMaxLikModel :426-473 (incl. scipy gaussian_kde).
//
// One 256-thread workgroup per sample x locus unit.  The grid rectangle h1range x h2range is walked
// pair-per-thread; every pair's log-likelihood is a short sum over the unit's sparse observations
// (distinct FULL sizes, distinct PREF/POST sizes, spanning-pair lengths), so only the entries of the
// reference's dense 1000-vectors that are actually read are ever computed.  Reductions (arg-max
// with the reference's tie-break, PP sums) go through wave shuffles + LDS partials; marginals are
// accumulated in the reference's own enumeration order so the CI walk sees the same sums.
//
// Arithmetic mirrors the reference's operation order (compiled with -ffp-contract=off); the only
// intended differences are libm-vs-ocml last-bit effects in log/exp/lgamma.
#include "tredgpu_internal.h"

namespace tredgpu {
namespace {

constexpr int NT = 128;   // threads per workgroup (one unit at a time); 6 workgroups per CU
constexpr int XPER = (TREDGPU_SPAN + NT - 1) / NT;  // KDE x-values per thread
constexpr int SPAN = TREDGPU_SPAN;
constexpr int MAXOBS = 256;  // distinct FULL / PREF sizes per unit
constexpr int MAXM = 768;    // marginal bins (repeat units)

struct Axis {
    int nb;      // entries taken from the sorted base list
    int start;   // arithmetic part: start, start+period, ...
    int n;       // its length
    __device__ int size() const { return nb + n; }
};

struct Obs {
    int fullK[MAXOBS], fullC[MAXOBS];
    int partK[MAXOBS], partC[MAXOBS];
    int base[MAXOBS + 1];
    int nF, nP, nb;
};

__device__ __forceinline__ int axis_value(const Axis& ax, const int* base, int period, int idx) {
    return idx < ax.nb ? base[idx] : ax.start + (idx - ax.nb) * period;
}

// NoiseModel.predict, models.py:79-84 with x = (period, h/period, gc, score), :156
__device__ double stutter_prob(const ModelConst& M, int period, int h) {
    double z = M.w[0];
    z += M.w[1] * (double)period;
    z += M.w[2] * (double)(h / period);
    z += M.w[3] * M.gc;
    z += M.w[4] * M.score;
    return 1.0 / (1 + exp(-1 * z));
}

// entry k of pdf_spanning(h), models.py:149-168 (right-aligned slice quirk at the array end kept)
__device__ double spanning_at(const double* step, double pi, int h, int k) {
    int start = h - 18, end = h + 19;
    if (start < 0) start = 0;
    if (end > SPAN) end = SPAN;
    if (k < start || k >= end) return 0.0;
    const int idx = 37 - (end - start) + (k - start);
    return idx == 18 ? 1 - pi : step[idx] * pi;
}

// entry k of pdf_partial(h), models.py:170-180; hp = min(h, max_partial), pi = stutter_prob(hp)
__device__ double partial_at(const double* step, double pi_hp, int hp, int k) {
    const double c = 1. / (hp + 1);
    double a = k < hp ? c : 0.0;
    a += c * spanning_at(step, pi_hp, hp, k);
    return a;
}

// entry x of PEMaxLikModel.roll(h), models.py:441-458
__device__ double roll_at(const double* pdf, int ref_len, int minpe, int h, int x, double small) {
    const int shift = ref_len - h;
    int src = (x - shift) % SPAN;
    if (src < 0) src += SPAN;
    double p = pdf[src];
    if (shift > 0) {
        if (x < shift) p = small;
    } else if (shift < 0) {
        const int from = shift < -SPAN ? 0 : SPAN + shift;
        if (x >= from) p = small;
    }
    if (x < minpe) p = small;
    return p;
}

struct PairCtx {
    const ModelConst* M;
    const double* step;  // step-size row of this period
    const Obs* obs;
    const double* pdf;   // KDE (LDS) when run_pe
    const int32_t* tl;   // target lens of the unit
    int n_target;
    int period, readlen, t1, t2, mp_eff, ref_len, minpe, n_rept;
    bool run_pe;
    double half_depth, lgam_rept, small, really_small, logsmall;
};

// log(max(p, SMALL_VALUE)) (safe_log, models.py:418-423); log(SMALL_VALUE) itself is evaluated once
// per unit with the same device log, so clamped terms cost no transcendental.
__device__ __forceinline__ double safe_log(const PairCtx& C, double p) {
    if (p < C.small) return C.logsmall;
    return log(p);
}

// spanning + partial terms (models.py:192-207)
__device__ void eval_reads(const PairCtx& C, int h1, int h2, double& ml1, double& ml2) {
    const Obs& O = *C.obs;
    ml1 = 0;
    if (O.nF > 0) {
        const double pi1 = stutter_prob(*C.M, C.period, h1), pi2 = stutter_prob(*C.M, C.period, h2);
        const int s1 = max(0, C.t2 - h1), s2 = max(0, C.t2 - h2);
        const double alpha = (s1 + s2) ? s1 * 1. / (s1 + s2) : .5;
        for (int i = 0; i < O.nF; ++i) {
            const int k = O.fullK[i];
            const double p = alpha * spanning_at(C.step, pi1, h1, k) + (1 - alpha) * spanning_at(C.step, pi2, h2, k);
            ml1 += safe_log(C, p) * O.fullC[i];
        }
    }
    ml2 = 0;
    if (O.nP > 0) {
        const int hp1 = min(h1, C.mp_eff), hp2 = min(h2, C.mp_eff);
        const double pi1 = stutter_prob(*C.M, C.period, hp1), pi2 = stutter_prob(*C.M, C.period, hp2);
        const int s1 = min(h1, C.t1), s2 = min(h2, C.t1);
        const double alpha = (s1 + s2) ? s1 * 1. / (s1 + s2) : .5;
        for (int i = 0; i < O.nP; ++i) {
            const int k = O.partK[i];
            const double p = alpha * partial_at(C.step, pi1, hp1, k) + (1 - alpha) * partial_at(C.step, pi2, hp2, k);
            ml2 += safe_log(C, p) * O.partC[i];
        }
    }
}

// repeat-only term (models.py:209-221); a function of dsum = max(h1-L,1) + max(h2-L,1) only.
// scipy poisson.pmf = exp(xlogy(k,mu) - gammaln(k+1) - mu)
__device__ double rept_term(const PairCtx& C, int dsum) {
    const double mu = dsum * C.half_depth / C.readlen;
    const double xl = C.n_rept == 0 ? 0.0 : C.n_rept * log(mu);
    double prob = exp(xl - C.lgam_rept - mu);
    if (!(prob > C.really_small)) prob = C.really_small;
    return log(prob);
}

// paired-end term (models.py:460-473).  r1/r2: roll(h1)[x_t], roll(h2)[x_t] for the unit's spanning
// pairs, either from the per-axis tables or evaluated on the fly (same expression, same bits).
// Sum of log(max(p, SMALL)) evaluated as the log of a running product that is flushed before it can
// leave the normal range (every factor is in [e^-10, 1], so 32 factors stay above e^-320): one log per
// 32 pairs.  Differs from the reference's term-by-term sum by O(1e-14), far inside the 1e-6 contract.
template <bool TABLES>
__device__ double pe_term(const PairCtx& C, int h1, int h2, const double* r1, const double* r2, int r2_stride) {
    double ml4 = 0, prod = 1.0;
    const int n = C.n_target;
    if (TABLES) {
        // 8 factors per step with all 16 loads issued up front (the tables live in L2-resident scratch);
        // same multiplication order as the plain loop, padding factors are exactly 1.0
        for (int base = 0; base < n; base += 8) {
            double a[8], b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = min(base + q, n - 1);
                a[q] = r1[idx];
                b[q] = r2[(size_t)idx * r2_stride];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double p = .5 * a[q] + (1 - .5) * b[q];
                if (p < C.small) p = C.small;
                if (base + q >= n) p = 1.0;
                prod *= p;
            }
            if ((base & 31) == 24) { ml4 += log(prod); prod = 1.0; }
        }
        if (n & 31) ml4 += log(prod);
        return ml4;
    }
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int x = C.tl[i];
        if (x < 0) x += SPAN;
        const double p1 = roll_at(C.pdf, C.ref_len, C.minpe, h1, x, C.small);
        const double p2 = roll_at(C.pdf, C.ref_len, C.minpe, h2, x, C.small);
        double p = .5 * p1 + (1 - .5) * p2;
        if (p < C.small) p = C.small;
        prod *= p;
        if (++k == 32) { ml4 += log(prod); prod = 1.0; k = 0; }
    }
    if (k) ml4 += log(prod);
    return ml4;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum; result valid in every thread
__device__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    return t;
}

// gaussian_kde(global_lens).evaluate(arange(1000)) / sum  (models.py:428-435; scipy: Scott factor
// n^(-1/5), covariance with ddof=1, kernel exp(-((l-x)/sigma)^2/2) / (sigma*sqrt(2*pi)) / n).
// hist/kern are LDS scratch of SPAN ints / SPAN doubles; pdf receives the result (LDS or global).
// Returns 0, or -2 (singular / too few points), -6 (length outside [0,1000)).
__device__ int kde_block(const int32_t* lens, int n, int* hist, double* kern, double* pdf, double* red, int* flag) {
    const int tid = threadIdx.x;
    for (int i = tid; i < SPAN; i += NT) hist[i] = 0;
    if (tid == 0) *flag = 0;
    __syncthreads();
    double s = 0;
    for (int i = tid; i < n; i += NT) {
        const int v = lens[i];
        if (v < 0 || v >= SPAN) atomicOr(flag, 1);
        else atomicAdd(&hist[v], 1);
        s += (double)v;
    }
    const double total = block_sum(s, red);
    if (*flag || n >= 65536) return -6;
    if (n < 2) return -2;
    const double mean = total / n;
    double q = 0;
    for (int i = tid; i < SPAN; i += NT) {
        const double d = (double)i - mean;
        q += hist[i] * (d * d);
    }
    const double var = block_sum(q, red) / (n - 1);
    if (!(var > 0)) return -2;
    const double factor = pow((double)n, -1. / 5);
    const double sigma = sqrt(var) * factor;       // cho_cov
    const double norm = 1.0 / sqrt(2 * M_PI) / sigma;  // (2*pi)^(-d/2) / cho_cov
    const double w = 1.0 / n;                      // uniform weights
    // kern[d] = exp(-(d/sigma)^2 / 2) * norm, d = |l - x|
    for (int d = tid; d < SPAN; d += NT) {
        const double r = (double)d / sigma;
        kern[d] = exp(-(r * r) / 2) * norm;
    }
    __syncthreads();
    // compact the non-empty bins in place (ascending v, as the dense walk would visit them):
    // hist[k] = v<<16 | count for k < nnz.  One wave, ballot + prefix popcount, 64 bins per step.
    if (tid < 64) {
        int nnz = 0;
        for (int base = 0; base < SPAN; base += 64) {
            const int v = base + tid;
            const int c = v < SPAN ? hist[v] : 0;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(c > 0);
            const int before = __builtin_popcountll(mask & ((1ull << tid) - 1ull));
            // all reads of this 64-bin group happened above; writes go to indices <= base + tid
            if (c > 0) hist[nnz + before] = (v << 16) | c;
            nnz += __builtin_popcountll(mask);
        }
        if (tid == 0) *flag = nnz;
    }
    __syncthreads();
    const int nnz = *flag;
    // every thread owns x = tid, tid+NT, ...: the bin list is walked once for all of them
    double acc[XPER];
#pragma unroll
    for (int q = 0; q < XPER; ++q) acc[q] = 0;
    for (int k = 0; k < nnz; ++k) {
        const int e = hist[k];
        const int v = e >> 16;
        const double wk = (e & 0xFFFF) * w;
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int x = tid + q * NT;
            if (x < SPAN) acc[q] += wk * kern[x > v ? x - v : v - x];
        }
    }
    double part = 0;
#pragma unroll
    for (int q = 0; q < XPER; ++q) {
        const int x = tid + q * NT;
        if (x < SPAN) { pdf[x] = acc[q]; part += acc[q]; }
    }
    const double tot = block_sum(part, red);
    for (int x = tid; x < SPAN; x += NT) pdf[x] = pdf[x] / tot;
    __syncthreads();
    return 0;
}

__global__ __launch_bounds__(NT) void pe_kde_kernel(GridArgs a) {
    __shared__ int hist[SPAN];
    __shared__ double kern[SPAN];
    __shared__ double red[NT / 64];
    __shared__ int flag;
    const int g = blockIdx.x;
    const tredgpu_unit_params u = a.units[g];
    const int rc = kde_block(a.global_lens + u.pe_off, u.n_global, hist, kern, a.kde_pdf + (size_t)g * SPAN, red, &flag);
    if (threadIdx.x == 0) a.kde_status[g] = rc;
}

struct Best {
    double ml;
    int h1, pos;
};
__device__ __forceinline__ bool better(const Best& x, const Best& y) {  // is x preferred over y
    if (x.pos < 0) return false;
    if (y.pos < 0) return true;
    if (x.ml != y.ml) return x.ml > y.ml;
    if (x.h1 != y.h1) return x.h1 < y.h1;   // key (ml, -h1), models.py:299
    return x.pos < y.pos;                   // python max keeps the first maximal element
}

constexpr int OBSMAX = MAXOBS;

struct GridShared {
    Obs obs;
    union {
        int hist[SPAN];               // KDE scratch / raw histograms while loading
        int row_off[GRID_MAX_ROWS + 1];
    };
    union {
        double kern[SPAN];            // KDE scratch ...
        double ph1[MAXM];             // ... then the P_h1 marginal (after pass B)
    };
    union {
        double pdf[SPAN];             // normalised KDE (read until the end of pass A) ...
        double ph2[MAXM];             // ... then the P_h2 marginal
    };
    double red[NT / 64];
    Best bred[NT / 64];
    int flag;
    int status;
    int unit;
};
static_assert(MAXM <= SPAN, "marginals alias the KDE arrays");

__global__ __launch_bounds__(NT, 3) void grid_kernel(GridArgs a, double* scratch, int* next_unit) {
    constexpr size_t scratch_per_block = GRID_SCRATCH_DOUBLES;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    GridShared& S = *reinterpret_cast<GridShared*>(smem_raw);
    const int tid = threadIdx.x;
    // per-workgroup scratch (global, L2-resident): [rept table][roll table rows][roll table cols][ml grid]
    double* const rept_tab = scratch + (size_t)blockIdx.x * scratch_per_block;
    double* const roll1 = rept_tab + GRID_REPT_TAB;
    double* const roll2 = roll1 + (size_t)GRID_MAX_ROWS * GRID_TMAX;
    double* const far1 = roll2 + (size_t)GRID_MAX_COLS * GRID_TMAX;   // per-row ml1 / ml2 against any "far" h2
    double* const far2 = far1 + GRID_MAX_ROWS;
    double* const mlbuf = far2 + GRID_MAX_ROWS;
    const ModelConst& M = *a.model;

    while (true) {
        // dynamic unit scheduling: grids differ by three orders of magnitude in size
        __syncthreads();
        if (tid == 0) S.unit = atomicAdd(next_unit, 1);
        __syncthreads();
        const int g = S.unit;
        if (g >= a.n_units) break;
        const tredgpu_unit_params u = a.units[g];
        const int period = u.period, readlen = u.readlen;
        const int t1 = readlen - 9, t2 = readlen - 18, t3 = readlen - 27;  // models.py:114-116
        tredgpu_call call;
        call.status = 0; call.n_pairs = 0; call.h1 = call.h2 = -1;
        call.ci[0] = call.ci[1] = call.ci[2] = call.ci[3] = 0;
        call.run_pe = 0; call.pad = 0; call.lik = -1; call.pp = -1;

        // ---- observations (models.py:399-403): sparse lists in bp, ascending ----
        const bool staged = 3 * a.hist_stride <= SPAN;
        if (staged) {
            for (int h = tid; h < a.hist_stride; h += NT) {
                S.hist[h] = a.full_cnt[(size_t)g * a.hist_stride + h];
                S.hist[a.hist_stride + h] = a.pref_cnt[(size_t)g * a.hist_stride + h];
                S.hist[2 * a.hist_stride + h] = a.rept_cnt[(size_t)g * a.hist_stride + h];
            }
            __syncthreads();
        }
        if (tid == 0) {
            int nF = 0, nP = 0, rept = 0, st = 0;
            const int32_t* fc = staged ? S.hist : a.full_cnt + (size_t)g * a.hist_stride;
            const int32_t* pc = staged ? S.hist + a.hist_stride : a.pref_cnt + (size_t)g * a.hist_stride;
            const int32_t* rc = staged ? S.hist + 2 * a.hist_stride : a.rept_cnt + (size_t)g * a.hist_stride;
            for (int h = 0; h < a.hist_stride; ++h) {
                if (fc[h] > 0) { if (nF < OBSMAX) { S.obs.fullK[nF] = h * period; S.obs.fullC[nF] = fc[h]; } ++nF; }
                if (pc[h] > 0) { if (nP < OBSMAX) { S.obs.partK[nP] = h * period; S.obs.partC[nP] = pc[h]; } ++nP; }
                rept += rc[h];
            }
            if (nF > OBSMAX || nP > OBSMAX) st = -9;
            if (period < 1 || period >= 18) st = -7;  // step_size_by_period KeyError, models.py:157
            S.obs.nF = min(nF, OBSMAX);
            S.obs.nP = min(nP, OBSMAX);
            S.flag = rept;
            S.status = st;
        }
        __syncthreads();
        const int nF = S.obs.nF, nP = S.obs.nP, n_rept = S.flag;
        int status = S.status;
        const int max_full = nF ? S.obs.fullK[nF - 1] : 0;
        const int max_partial = nP ? S.obs.partK[nP - 1] : 0;
        int reads_above_full = 0;
        for (int i = 0; i < nP; ++i)
            if (S.obs.partK[i] > max_full + period) reads_above_full += S.obs.partC[i];
        // observation sizes index the 1000-vectors (models.py:198,206): IndexError past the end
        if (status == 0 && (max_full >= SPAN || max_partial >= SPAN)) status = -3;
        __syncthreads();  // S.hist is reused by the KDE below

        // ---- paired-end model (models.py:131-132, 428-439) ----
        const bool have_pe = u.n_global >= 100 && u.n_target >= 5;
        const bool run_pe = max_partial >= t3 && reads_above_full > 1 && have_pe;  // :234-236
        if (status == 0 && have_pe) {
            // the reference builds the KDE whenever the model exists; a singular one raises there
            int rc;
            if (!run_pe) {
                // only the singularity check matters: all lengths equal <=> zero variance
                double sum = 0;
                const int32_t* gl = a.global_lens + u.pe_off;
                for (int i = tid; i < u.n_global; i += NT) sum += (double)gl[i];
                const double mean = block_sum(sum, S.red) / u.n_global;
                double q = 0;
                for (int i = tid; i < u.n_global; i += NT) { const double d = gl[i] - mean; q += d * d; }
                rc = block_sum(q, S.red) > 0 ? 0 : -2;
            } else {
                rc = kde_block(a.global_lens + u.pe_off, u.n_global, S.hist, S.kern, S.pdf, S.red, &S.flag);
            }
            if (rc) status = rc;
        }
        if (status == 0 && run_pe) {
            for (int i = 0; i < u.n_target; ++i) {
                int x = a.target_lens[u.tl_off + i];
                if (x < 0) x += SPAN;
                if (x < 0 || x >= SPAN) status = -3;
            }
        }

        // ---- grid axes (models.py:239-257) ----
        if (tid == 0) {
            int nb = 0, i = 0;
            bool mp_done = nP == 0;
            while (i < nF || !mp_done) {  // sorted(set(FULL keys) | {max_partial})
                int v;
                if (i < nF && (mp_done || S.obs.fullK[i] <= max_partial)) {
                    v = S.obs.fullK[i++];
                    if (!mp_done && v == max_partial) mp_done = true;
                } else { v = max_partial; mp_done = true; }
                S.obs.base[nb++] = v;
            }
            S.obs.nb = nb;
        }
        __syncthreads();
        const int nb = S.obs.nb;
        if (status == 0 && nb == 0) status = 1;  // no evidence: alleles (-1,-1), models.py:244-245
        const int mp_eff = max(t2, max_partial);  // self.max_partial, models.py:117,241-242
        Axis ext, bas, ful, ax1, ax2;
        bas.nb = nb; bas.start = 0; bas.n = 0;
        ext.nb = nb; ext.start = max_partial + period;
        ext.n = period * u.maxinsert + 1 > ext.start ? (period * u.maxinsert - ext.start) / period + 1 : 0;
        ful.nb = 0; ful.start = period; ful.n = u.maxinsert > 0 ? u.maxinsert : 0;
        if (u.fullsearch) { ax1 = ful; ax2 = ful; }
        else {
            ax1 = max_full ? bas : ext;
            ax2 = (n_rept || run_pe) ? ext : bas;
        }
        const int nrow = ax1.size();
        const int ncol = u.ploidy == 1 ? 1 : ax2.size();
        if (status == 0 && (nrow > GRID_MAX_ROWS || ncol > GRID_MAX_COLS)) status = -5;
        if (status == 0 && (nrow == 0 || ncol == 0)) status = -8;
        if (status == 0) {   // marginals are indexed by repeat units: every axis value must fit
            int hm = axis_value(ax1, S.obs.base, period, nrow - 1);
            if (u.ploidy != 1) hm = max(hm, axis_value(ax2, S.obs.base, period, ncol - 1));
            if (nb > 0) hm = max(hm, S.obs.base[nb - 1]);
            if (hm / period >= MAXM) status = -5;
        }

        if (status != 0) {
            if (tid == 0) { call.status = status; call.run_pe = run_pe; a.calls[g] = call; }
            if (a.marg != nullptr)
                for (int m = tid; m < 2 * a.marg_stride; m += NT) a.marg[(size_t)g * 2 * a.marg_stride + m] = 0;
            continue;
        }

        PairCtx C;
        C.M = &M;
        C.step = M.step[period <= 6 ? period - 1 : 5];  // models.py:54-60
        C.obs = &S.obs;
        C.pdf = S.pdf;
        C.tl = a.target_lens + u.tl_off;
        C.n_target = u.n_target;
        C.period = period; C.readlen = readlen; C.t1 = t1; C.t2 = t2; C.mp_eff = mp_eff;
        C.ref_len = u.ref_len; C.minpe = u.minpe; C.n_rept = n_rept; C.run_pe = run_pe;
        C.half_depth = u.half_depth;
        C.lgam_rept = lgamma((double)n_rept + 1);
        C.small = M.small; C.really_small = M.really_small;
        C.logsmall = log(M.small);

        // ---- rows: count of valid h2 per h1 (h1 <= h2), dump offsets; per-row "far" terms ----
        // For h2 >= h_far the spanning and partial terms no longer depend on h2 (S(k|h2) = 0 for every
        // observed size, alpha is pinned, pdf_partial is clipped at max_partial): bit-identical values,
        // evaluated once per row instead of once per pair.
        const int h_far = max(max(max_full + 19, mp_eff), t1);
        for (int i = tid; i < nrow; i += NT) {
            const int h1 = axis_value(ax1, S.obs.base, period, i);
            int cnt = 0;
            if (u.ploidy == 1) cnt = 1;
            else for (int j = 0; j < ncol; ++j) cnt += axis_value(ax2, S.obs.base, period, j) >= h1;
            S.row_off[i] = cnt;
            eval_reads(C, h1, max(h_far, h1), far1[i], far2[i]);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int i = 0; i < nrow; ++i) { const int c = S.row_off[i]; S.row_off[i] = acc; acc += c; }
            S.row_off[nrow] = acc;
        }
        __syncthreads();
        const int n_pairs = S.row_off[nrow];
        int64_t dump_base = -1;
        if (a.grid_dump != nullptr) {
            const int64_t cap = a.grid_off[g + 1] - a.grid_off[g];
            if (n_pairs <= cap) dump_base = a.grid_off[g];
        }

        // ---- tables for big grids: the repeat-only term depends on dsum only, the paired-end term on
        //      roll(h)[x_t] per axis value; both are filled with the very expressions the direct path uses
        const int rect = nrow * ncol;
        const int last1 = axis_value(ax1, S.obs.base, period, nrow - 1);
        const int last2 = u.ploidy == 1 ? last1 : axis_value(ax2, S.obs.base, period, ncol - 1);
        int hmaxv = max(last1, last2);
        if (nb > 0) hmaxv = max(hmaxv, S.obs.base[nb - 1]);
        const int dmax = 2 * max(hmaxv - readlen, 1);
        const bool use_rept_tab = rect >= 1024 && dmax < GRID_REPT_TAB;
        const bool use_roll_tab = run_pe && rect >= 1024 && u.n_target <= GRID_TMAX;
        if (use_rept_tab) {
            // only the dsum values that occur: {2} U {1 + d} U {d + d'} for d, d' in D = {h - L > 0}
            for (int d = 2 + tid; d <= dmax; d += NT) rept_tab[d] = 1.0;   // 1.0 = unset (terms are <= 0)
            __syncthreads();
            int i = tid / ncol, j = tid - i * ncol;
            for (int pos = tid; pos < rect; pos += NT) {
                const int h1 = axis_value(ax1, S.obs.base, period, i);
                const int h2 = u.ploidy == 1 ? h1 : axis_value(ax2, S.obs.base, period, j);
                if (h1 <= h2) rept_tab[max(h1 - readlen, 1) + max(h2 - readlen, 1)] = 2.0;  // needed
                j += NT;
                while (j >= ncol) { j -= ncol; ++i; }
            }
            __syncthreads();
            for (int d = 2 + tid; d <= dmax; d += NT)
                if (rept_tab[d] == 2.0) rept_tab[d] = rept_term(C, d);
        }
        if (use_roll_tab) {
            const int nt = u.n_target;
            for (int k = tid; k < (nrow + ncol) * nt; k += NT) {
                const bool isrow = k < nrow * nt;
                const int kk = isrow ? k : k - nrow * nt;
                const int ai = kk / nt, t = kk - ai * nt;
                const int h = isrow ? axis_value(ax1, S.obs.base, period, ai)
                                    : (u.ploidy == 1 ? 0 : axis_value(ax2, S.obs.base, period, ai));
                int x = C.tl[t];
                if (x < 0) x += SPAN;
                const double rv = roll_at(C.pdf, C.ref_len, C.minpe, h, x, C.small);
                if (isrow) roll1[(size_t)ai * nt + t] = rv;
                else roll2[(size_t)t * ncol + ai] = rv;   // transposed: coalesced across columns in pass A
            }
        }
        __syncthreads();

        // ---- pass A: log-likelihood of every pair, arg-max ----
        Best mine; mine.ml = 0; mine.h1 = 0; mine.pos = -1;
        {
            int i = tid / ncol, j = tid - i * ncol;   // (row, column) of pos, advanced incrementally
            for (int pos = tid; pos < rect; pos += NT) {
                const int h1 = axis_value(ax1, S.obs.base, period, i);
                const int h2 = u.ploidy == 1 ? h1 : axis_value(ax2, S.obs.base, period, j);
                if (h1 > h2) mlbuf[pos] = 0;
                else {
                    double ml4[4];
                    if (h2 >= h_far) { ml4[0] = far1[i]; ml4[1] = far2[i]; }
                    else eval_reads(C, h1, h2, ml4[0], ml4[1]);
                    const int dsum = max(h1 - readlen, 1) + max(h2 - readlen, 1);
                    ml4[2] = use_rept_tab ? rept_tab[dsum] : rept_term(C, dsum);
                    ml4[3] = 0;
                    if (run_pe) {
                        if (use_roll_tab && u.ploidy != 1)
                            ml4[3] = pe_term<true>(C, h1, h2, roll1 + (size_t)i * u.n_target, roll2 + j, ncol);
                        else ml4[3] = pe_term<false>(C, h1, h2, nullptr, nullptr, 0);
                    }
                    const double ml = ml4[0] + ml4[1] + ml4[2] + ml4[3];  // models.py:269
                    mlbuf[pos] = ml;
                    Best b; b.ml = ml; b.h1 = h1; b.pos = pos;
                    if (better(b, mine)) mine = b;
                    if (dump_base >= 0) {
                        int within = 0;  // valid columns before j in this row
                        if (u.ploidy != 1)
                            for (int jj = 0; jj < j; ++jj) within += axis_value(ax2, S.obs.base, period, jj) >= h1;
                        double* d = a.grid_dump + (dump_base + S.row_off[i] + within) * 6;
                        d[0] = h1; d[1] = h2; d[2] = ml4[0]; d[3] = ml4[1]; d[4] = ml4[2]; d[5] = ml4[3];
                    }
                }
                j += NT;
                while (j >= ncol) { j -= ncol; ++i; }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            Best other;
            other.ml = __shfl_down(mine.ml, o, 64);
            other.h1 = __shfl_down(mine.h1, o, 64);
            other.pos = __shfl_down(mine.pos, o, 64);
            if (better(other, mine)) mine = other;
        }
        if ((tid & 63) == 0) S.bred[tid >> 6] = mine;
        __syncthreads();
        Best top = S.bred[0];
        for (int w = 1; w < NT / 64; ++w) if (better(S.bred[w], top)) top = S.bred[w];
        const double max_ml = top.ml;

        // ---- pass B: exp(ml - max) once per pair (models.py:280-285) + the PP sums (:342-368) ----
        double all = 0, path = 0;
        {
            int i = tid / ncol, j = tid - i * ncol;
            for (int pos = tid; pos < rect; pos += NT) {
                const int h1 = axis_value(ax1, S.obs.base, period, i);
                const int h2 = u.ploidy == 1 ? h1 : axis_value(ax2, S.obs.base, period, j);
                if (h1 <= h2) {   // else mlbuf[pos] stays 0: contributes nothing to the sums below
                    const double e = exp(mlbuf[pos] - max_ml);
                    mlbuf[pos] = e;
                    all += e;
                    const int lo = h1 / period, hi = h2 / period;
                    bool p;
                    if (u.is_expansion) p = (u.is_recessive ? lo : hi) >= u.cutoff_risk;
                    else p = (u.is_recessive ? hi : lo) <= u.cutoff_risk;
                    if (p) path += e;
                }
                j += NT;
                while (j >= ncol) { j -= ncol; ++i; }
            }
        }
        all = block_sum(all, S.red);
        path = block_sum(path, S.red);
        const int mlim = min(MAXM, hmaxv / period + 1);
        for (int m = tid; m < mlim; m += NT) { S.ph1[m] = 0; S.ph2[m] = 0; }
        __syncthreads();  // also orders the mlbuf writes above before the reads below
        // marginal P_h1: one wave per row (fixed shuffle tree), rows merged by key in row order
        for (int i = tid >> 6; i < nrow; i += NT / 64) {
            double acc = 0;
            for (int j = tid & 63; j < ncol; j += 64) acc += mlbuf[i * ncol + j];
            acc = wave_sum(acc);
            if ((tid & 63) == 0) far1[i] = acc;  // far1 is free now: row sums
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < nrow; ++i) {
                const int m = axis_value(ax1, S.obs.base, period, i) / period;
                if (m < MAXM) S.ph1[m] += far1[i];
            }
        }
        // marginal P_h2: one thread per distinct h2 value, rows outermost as in the reference.  The
        // extended axis can list a value twice (base part + arithmetic part, models.py:251-252): the
        // first occurrence owns the sum.
        if (u.ploidy != 1) {
            for (int j = tid; j < ncol; j += NT) {
                const int h2 = axis_value(ax2, S.obs.base, period, j);
                int twin = -1;
                if (j < ax2.nb) {
                    const int d = h2 - ax2.start;
                    if (ax2.n > 0 && d >= 0 && d % period == 0 && d / period < ax2.n) twin = ax2.nb + d / period;
                } else {
                    bool dup = false;
                    for (int k = 0; k < ax2.nb; ++k) dup |= S.obs.base[k] == h2;
                    if (dup) continue;  // owned by the base occurrence
                }
                double acc = 0;
                for (int i = 0; i < nrow; ++i) {
                    acc += mlbuf[i * ncol + j];
                    if (twin >= 0) acc += mlbuf[i * ncol + twin];
                }
                const int m = h2 / period;
                if (m < MAXM) S.ph2[m] = acc;
            }
        }
        __syncthreads();
        if (u.ploidy == 1) {
            for (int m = tid; m < mlim; m += NT) S.ph2[m] = S.ph1[m];  // h2 == h1 for every pair
            __syncthreads();
        }

        if (tid == 0) {
            // calc_CI, models.py:319-340 on each marginal
            for (int which = 0; which < 2; ++which) {
                const double* P = which ? S.ph2 : S.ph1;
                double total = 0;
                for (int m = 0; m < mlim; ++m) total += P[m];
                double cum = 0;
                int lo = 0, hi = 0, last = 0;
                bool in_range = false, broke = false;
                for (int m = 0; m < mlim && !broke; ++m) {
                    if (P[m] == 0) continue;
                    last = m;
                    cum += P[m];
                    if (!in_range && cum > .025 * total) { in_range = true; lo = m; }
                    if (cum > .975 * total) broke = true;
                }
                hi = last;
                call.ci[2 * which] = lo;
                call.ci[2 * which + 1] = hi;
            }
            const int bi = top.pos / ncol, bj = top.pos - bi * ncol;
            call.h1 = axis_value(ax1, S.obs.base, period, bi);
            call.h2 = u.ploidy == 1 ? call.h1 : axis_value(ax2, S.obs.base, period, bj);
            call.lik = max_ml;
            const double pp = path / all;
            call.pp = pp < 1 ? pp : 1;
            call.n_pairs = n_pairs;
            call.run_pe = run_pe;
            call.status = (a.grid_dump != nullptr && dump_base < 0) ? -4 : 0;
            a.calls[g] = call;
        }
        if (a.marg != nullptr) {
            for (int m = tid; m < a.marg_stride; m += NT) {
                a.marg[((size_t)g * 2 + 0) * a.marg_stride + m] = m < mlim ? S.ph1[m] : 0;
                a.marg[((size_t)g * 2 + 1) * a.marg_stride + m] = m < mlim ? S.ph2[m] : 0;
            }
        }
    }
}

}  // namespace

hipError_t launch_pe_kde(const GridArgs& a, hipStream_t s) {
    if (a.n_units <= 0) return hipSuccess;
    pe_kde_kernel<<<a.n_units, NT, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_grid(const GridArgs& a, double* scratch, int* next_unit, hipStream_t s) {
    if (a.n_units <= 0) return hipSuccess;
    hipError_t e0 = hipMemsetAsync(next_unit, 0, sizeof(int), s);
    if (e0 != hipSuccess) return e0;
    const int blocks = a.n_units < GRID_MAX_BLOCKS ? a.n_units : GRID_MAX_BLOCKS;
    const size_t smem = sizeof(GridShared);
    hipError_t e = hipFuncSetAttribute((const void*)grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    grid_kernel<<<blocks, NT, smem, s>>>(a, scratch, next_unit);
    return hipGetLastError();
}

}  // namespace tredgpu
