// grid.hip -- (h1,h2) allele-pair likelihood grid + paired-end KDE for gfx950 (fp64).
//
// Replaces /root/reference/tredparse/models.py:
//   pdf_spanning :149-168, pdf_partial :170-180, get_alpha :182-190, evaluate_spanning :192-198,
//   evaluate_partial :200-207, evaluate_rept :209-221, evaluate :223-302, calc_CI :319-340,
//   calc_PP :342-368, safe_log :418-423, PEMaxLikModel :426-473 (incl. scipy gaussian_kde).
//
// Four kernels per pass over the batch's units; every unit takes exactly the scratch it needs from one pool
// (atomic bump allocation in 16 sub-pools; a unit that finds its sub-pool full is deferred to a further pass):
//   grid_kde_kernel      one workgroup per unit that has a paired-end model: is the term used (from the histograms),
//                        then the KDE -- an fp64 convolution -- or, when it is not, only the singularity check
//                        (first pass only)
//   grid_prepare_kernel  one workgroup per unit: sparse observation lists, grid axes, per-unit tables, per-row
//                        "far" terms -> UnitDesc + scratch
//   grid_pairs_kernel    one wavefront per work item = (unit, 64 columns, <= 128 rows), lane = column: the
//                        log-likelihood of each pair -- a short sum over the unit's sparse observations, so only
//                        the entries of the reference's dense 1000-vectors that are actually read are ever
//                        computed -- and the item's arg-max
//   grid_reduce_kernel   one workgroup per unit: arg-max with the reference's tie-break over the items, then one
//                        pass over the grid: exp(ml - max), PP sums, marginals, CI
// (splitting keeps every kernel's register footprint at what its phase needs; the single-kernel version of
// this path needed 168 VGPRs plus 400 B of spills per lane and ran at 15 ms against 9 ms per 30 000 units).
// Units and work items are handed out dynamically through ticket queues (GridCounters).  Device helpers that more than
// one kernel uses are __forceinline__: left to the compiler, kde_block and the block sums became real calls -- LDS
// through flat addresses, an ABI spill frame in the caller -- and the KDE ran at a third of its present speed.
//
// Arithmetic mirrors the reference's operation order (compiled with -ffp-contract=off); the only
// intended differences are libm-vs-ocml last-bit effects in log/exp, the paired-end product-log (pairs kernel) and
// the summation order of the marginals.
#include <algorithm>
#include <climits>
#include <type_traits>

#include "tredgpu_internal.h"

namespace tredgpu {
namespace {

constexpr int NT = 128;   // threads per workgroup (one unit at a time)
// waves per SIMD the two per-unit front kernels are compiled for, and the workgroups their launches keep resident
// (256 CUs x what registers and LDS admit per CU; further units come through the ticket queues)
#ifndef GRID_KDE_WAVES
#define GRID_KDE_WAVES 3
#endif
#ifndef GRID_PREP_WAVES
#define GRID_PREP_WAVES 4
#endif
#ifndef GRID_PAIRS_WAVES
#define GRID_PAIRS_WAVES 3
#endif
#ifndef GRID_REDUCE_WAVES
#define GRID_REDUCE_WAVES 4
#endif
constexpr int KDE_WAVES = GRID_KDE_WAVES, KDE_BLOCKS = 512 * GRID_KDE_WAVES;     // 160 VGPRs: the convolution's window registers
constexpr int PREP_WAVES = GRID_PREP_WAVES, PREP_BLOCKS = 512 * GRID_PREP_WAVES;   // 128 VGPRs (3 waves: 1.16 ms per 30 000 units, 4 waves: 1.02)
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
    const double* pdf;   // the unit's KDE (LDS copy of grid_kde_kernel's output) when run_pe
    const int32_t* tl;   // target lens of the unit
    int n_target;
    int period, readlen, t1, t2, mp_eff, ref_len, minpe, n_rept;
    bool run_pe;
    double half_depth, lgam_rept, small, really_small, logsmall;
};

// log(x) for positive, finite, normal x -- the two hot uses (a pair's product of <= 32 factors in [e^-10, 1]; an
// observation's probability >= SMALL_VALUE).  x = m * 2^e with m in [0.7071, 1.4142], s = (m - 1) / (m + 1),
// log(m) = 2 s + s z (2/3 + 2 z/5 + ... + 2 z^8/19), z = s^2 <= 0.0295 (the next term is < 3e-17 of the result);
// e ln2 is added in two parts.  < 2 ulp against a 120-bit reference over 7 M arguments (the device library's log
// is correctly rounded to the last bit or so and costs 89 instructions, 76 of them fp64; this is ~40, and the
// contract is 1e-6 absolute on sums of O(100) such terms).
__device__ __forceinline__ double pos_log(double x) {
    int hi = __double2hiint(x);
    const int lo = __double2loint(x);
    int e = (hi >> 20) - 1023;
    hi = (hi & 0x000FFFFF) | 0x3FF00000;                         // mantissa in [1, 2)
    double m = __hiloint2double(hi, lo);
    const bool up = m > 1.4142135623730951;
    m = up ? m * 0.5 : m;                                        // -> [0.7071, 1.4142]
    e += up ? 1 : 0;
    const double f = m - 1.0, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);                          // ~2^-23; two Newton steps
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-s, d, f), r, s);            // s = f / d to the last bit or so
    const double z = s * s;
    double p = 2.0 / 19;
    p = __builtin_fma(p, z, 2.0 / 17);
    p = __builtin_fma(p, z, 2.0 / 15);
    p = __builtin_fma(p, z, 2.0 / 13);
    p = __builtin_fma(p, z, 2.0 / 11);
    p = __builtin_fma(p, z, 2.0 / 9);
    p = __builtin_fma(p, z, 2.0 / 7);
    p = __builtin_fma(p, z, 2.0 / 5);
    p = __builtin_fma(p, z, 2.0 / 3);
    const double ef = (double)e;
    const double t = __builtin_fma(s * z, p, ef * 1.90821492927058770002e-10);
    return __builtin_fma(ef, 6.93147180369123816490e-01, 2.0 * s + t);
}

// log(max(p, SMALL_VALUE)) (safe_log, models.py:418-423); log(SMALL_VALUE) itself is evaluated once
// per unit with the same device log, so clamped terms cost no transcendental.
__device__ __forceinline__ double safe_log(const PairCtx& C, double p) {
    if (p < C.small) return C.logsmall;
    return pos_log(p);
}

// spanning + partial terms (models.py:192-207), evaluated by a group of G adjacent lanes (G a power of two <= 32, the
// same for all of them; lane g of the group takes observations g, g + G, ... and a butterfly adds the parts up, so
// every lane of the group returns the sums).  G = 1 is the plain serial sum in the reference's order; small grids use
// larger groups: with one thread per row a 20-row grid kept 20 of the workgroup's 128 lanes busy for 30 iterations.
__device__ void eval_reads(const PairCtx& C, int h1, int h2, double& ml1, double& ml2, int g = 0, int G = 1) {
    const Obs& O = *C.obs;
    ml1 = 0;
    if (O.nF > 0) {
        const double pi1 = stutter_prob(*C.M, C.period, h1), pi2 = stutter_prob(*C.M, C.period, h2);
        const int s1 = max(0, C.t2 - h1), s2 = max(0, C.t2 - h2);
        const double alpha = (s1 + s2) ? s1 * 1. / (s1 + s2) : .5;
        for (int i = g; i < O.nF; i += G) {
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
        for (int i = g; i < O.nP; i += G) {
            const int k = O.partK[i];
            const double p = alpha * partial_at(C.step, pi1, hp1, k) + (1 - alpha) * partial_at(C.step, pi2, hp2, k);
            ml2 += safe_log(C, p) * O.partC[i];
        }
    }
    for (int o = G >> 1; o > 0; o >>= 1) {
        ml1 += __shfl_xor(ml1, o, 64);
        ml2 += __shfl_xor(ml2, o, 64);
    }
}

// repeat-only term (models.py:209-221); a function of dsum = max(h1-L,1) + max(h2-L,1) only.
// scipy poisson.pmf = exp(xlogy(k,mu) - gammaln(k+1) - mu)
// (both logarithms through pos_log: its argument range -- positive, finite, normal -- is checked for mu and holds for
//  a probability in [e^-100, 1]; the device library's log cost half of the table's time, which was a quarter of
//  grid_prepare_kernel's)
__device__ __forceinline__ double rept_term(const PairCtx& C, int dsum) {
    const double mu = dsum * C.half_depth / C.readlen;
    double xl = 0.0;
    if (C.n_rept != 0) xl = C.n_rept * ((mu >= 2.3e-308 && mu <= 1.7e308) ? pos_log(mu) : log(mu));
    double prob = exp(xl - C.lgam_rept - mu);
    if (!(prob > C.really_small)) prob = C.really_small;
    return pos_log(prob);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum over the wavefront by DPP moves (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 and row_bcast:31):
// the total arrives in lane 63.  A fixed tree like wave_sum's, but ~100 cycles of latency instead of six LDS-crossbar
// round trips (ds_bpermute) -- it sits in the per-row chain of grid_reduce_kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return v + __hiloint2double(hi, lo);     // lanes without a source (or outside the row mask) add +0.0
}
template <bool MAX>
__device__ __forceinline__ int wave_minmax_to_last(int v) {
#define TRED_STEP(CTRL, ROW_MASK) { const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); v = MAX ? max(v, o) : min(v, o); }
    TRED_STEP(0x111, 0xF) TRED_STEP(0x112, 0xF) TRED_STEP(0x114, 0xF) TRED_STEP(0x118, 0xF) TRED_STEP(0x142, 0xA) TRED_STEP(0x143, 0xC)
#undef TRED_STEP
    return v;
}
__device__ __forceinline__ double wave_sum_to_last(double v) {
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// block-wide sum; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* red) {
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
// hist: SPAN ints, kern2: KERN2 doubles of LDS scratch; pdf receives the result (LDS or global).
// Returns 0, or -2 (singular / too few points), -6 (length outside [0,1000)).
//
// pdf[x] = sum_v count[v]/n * K(x - v), a convolution of the 1000-bin histogram with the Gaussian, cut at 9 sigma
// (see the loop).  K is even: it is stored for |d| = 0 .. 1039 at index swz(|d|) (i + i/8, the layout the
// histogram copy of the loop shares).
constexpr int KERN2_RAW = 1040;   // |d| <= NT * XPER - 1 + 7 = 1030 is the largest index a window touches
__device__ __forceinline__ int kswz(int i) { return i + (i >> 3); }
constexpr int KERN2 = KERN2_RAW + KERN2_RAW / 8 + 1;
constexpr int KHIST = SPAN + 2 + (SPAN + 2) / 8 + 1;   // the bins behind two zero guards, swizzled
static_assert(XPER == 8, "the sliding window below is unrolled for 8 x-values per thread");
static_assert(NT * XPER >= SPAN && NT * XPER + XPER <= KERN2_RAW && SPAN + XPER <= KERN2_RAW, "window indices stay inside kern2");

// First half: the histogram of the lengths (hist, zeroed by the caller behind a barrier), their sum, smallest and
// largest value; `bad` = some length outside [0, 1000).  The loads go out eight at a time (one after the other, each
// behind the LDS atomic of the one before, they were a chain of n / NT memory latencies per unit); the sum keeps the
// order of a plain strided walk.
struct KdeLens {
    double total;
    int lo, hi, bad;
};
__device__ __forceinline__ KdeLens kde_collect(const int32_t* lens, int n, int* hist, double* red, int* flag) {
    const int tid = threadIdx.x;
    double s = 0;
    int vlo = INT_MAX, vhi = INT_MIN, bad = 0;
    for (int base = tid; base < n; base += NT * 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = lens[min(base + k * NT, n - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (base + k * NT < n) {
                if (v[k] < 0 || v[k] >= SPAN) bad = 1;
                else atomicAdd(&hist[v[k]], 1);
                vlo = min(vlo, v[k]); vhi = max(vhi, v[k]);
                s += (double)v[k];
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        vlo = min(vlo, __shfl_down(vlo, o, 64)); vhi = max(vhi, __shfl_down(vhi, o, 64)); bad |= __shfl_down(bad, o, 64);
    }
    if ((tid & 63) == 0) { atomicMin(&flag[1], vlo); atomicMax(&flag[2], vhi); atomicOr(&flag[0], bad); }
    KdeLens r;
    r.total = block_sum(s, red);     // (its barriers publish the flags and the histogram)
    r.lo = flag[1]; r.hi = flag[2]; r.bad = flag[0];
    return r;
}
__device__ __forceinline__ void kde_clear(int* hist, int* flag) {   // the caller puts a barrier behind it
    for (int i = threadIdx.x; i < SPAN; i += NT) hist[i] = 0;
    if (threadIdx.x == 0) { flag[0] = 0; flag[1] = INT_MAX; flag[2] = INT_MIN; }
}

// All pair lengths equal to c: what the reference does then is scipy's business.  gaussian_kde takes the covariance from
// np.cov(data, aweights = 1/n each), i.e. from data - avg with avg = sum(c * w) / sum(w), both sums numpy's pairwise
// reductions (blocks of at most 128 elements on eight accumulators, halves cut at multiples of eight above that).  When
// that rounds to c exactly the covariance is 0 and the Cholesky factorisation raises LinAlgError (status -2); when it
// does not, the covariance is ~1e-28, the factorisation succeeds, and the normalised pdf over 0..999 is 1 at c and 0
// elsewhere.  Which of the two happens is a pure function of (c, n) -- 63 % of the pairs pass -- reproduced here step by
// step (checked against scipy 1.15.3 / numpy 2.2.6 on 8 239 pairs, tools/fuzz_hist.py draws more); one thread, an
// explicit stack for the halves (istack >= 48 ints, vstack >= 48 doubles).
__device__ double numpy_pairwise_sum_of_equal(double v, int n, int* istack, double* vstack) {
    int ni = 0, nv = 0;
    istack[ni++] = n;
    while (ni > 0) {
        const int m = istack[--ni];
        if (m < 0) {                                   // both halves of a block of -m are on the value stack
            const double right = vstack[--nv], left = vstack[--nv];
            vstack[nv++] = left + right;
        } else if (m < 8) {
            double r = 0.;
            for (int i = 0; i < m; ++i) r += v;
            vstack[nv++] = r;
        } else if (m <= 128) {
            double acc = v;                            // the eight accumulators hold the same number
            for (int i = 8; i < m - (m % 8); i += 8) acc += v;
            double r = ((acc + acc) + (acc + acc)) + ((acc + acc) + (acc + acc));
            for (int i = 0; i < m % 8; ++i) r += v;
            vstack[nv++] = r;
        } else {
            int n2 = m / 2;
            n2 -= n2 % 8;
            istack[ni++] = -m;
            istack[ni++] = m - n2;                     // (popped after the left half)
            istack[ni++] = n2;
        }
    }
    return vstack[0];
}
// 0: scipy builds the (one-hot) KDE of n lengths all equal to c; -2: it raises.  Called by every thread of the block;
// thread 0 works it out.
__device__ int kde_of_equal_lengths(int c, int n, int* istack, double* vstack, int* flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const double w = 1.0 / (double)n;
        const double avg = numpy_pairwise_sum_of_equal((double)c * w, n, istack, vstack) / numpy_pairwise_sum_of_equal(w, n, istack, vstack);
        flag[3] = ((double)c - avg) != 0.0 ? 0 : -2;
    }
    __syncthreads();
    return flag[3];
}

// Second half: from the histogram to the normalised pdf.
__device__ __forceinline__ int kde_finish(const KdeLens& in, int n, const int* hist, double* kern2, double* khist, double* pdf, double* red, int* flag) {
    const int tid = threadIdx.x;
    if (in.bad || n >= 65536) return -6;
    if (n < 2) return -2;
    if (in.lo == in.hi) {                              // (integers: the variance is zero exactly when smallest = largest)
        const int rc = kde_of_equal_lengths(in.lo, n, const_cast<int*>(hist), kern2, flag);
        if (rc == 0)
            for (int i = tid; i < SPAN; i += NT) pdf[i] = i == in.lo ? 1.0 : 0.0;
        return rc;
    }
    const double mean = in.total / n;
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
    // only bins within W = 9 sigma of x are added up (see the loop below); the window reads K(d) for |d| < W + 2 XPER
    const int W = (int)fmin((double)SPAN, ceil(9.0 * sigma));
    const int kend = W + 2 * XPER;
    // K(d) = exp(-(d/sigma)^2 / 2) * norm for |d| < 1000, 0 beyond
    for (int d = tid; d < min(SPAN, kend); d += NT) {
        const double r = (double)d / sigma;
        const double k = exp(-(r * r) / 2) * norm;
        kern2[kswz(d)] = k;
    }
    for (int i = SPAN + tid; i < min(KERN2_RAW, kend); i += NT) kern2[kswz(i)] = 0;
    // the bins once more, swizzled (i + i/8: the lanes below read with a stride of XPER ints); entry 0 and entry
    // SPAN + 1 are zero guards, bin v sits at v + 1: a lane's out-of-range bins clamp onto a guard, so the load in the
    // loop needs no branch
    // (as weights count / n, in double: the loop then is loads and fused multiply-adds only)
    for (int i = tid; i < SPAN + 2; i += NT) khist[kswz(i)] = (i >= 1 && i <= SPAN) ? hist[i - 1] * w : 0.0;
    __syncthreads();
    // A Gaussian term below 1e-17 of the kernel's peak cannot change a sum of at most 65 535 terms of that scale in
    // its 16th digit: only bins within W = 9 sigma of x are added up (exp(-81/2) = 2.6e-18).  Every thread owns XPER
    // consecutive x and walks d = v - x0 from -W to W + XPER - 1: the weight of bin x0 + d is a per-lane load, the
    // kernel values K(qx - d) are the same for all lanes and slide by one entry per step (rotating window of XPER
    // registers, one broadcast load per step).  With sigma ~ 17 bp (2 000 pairs, sd 80) that is 312 steps instead
    // of the ~850 of the occupied range of pair lengths (grid_prepare_kernel: 2.62 -> 2.38 ms per 30 000 units).
    // The products are accumulated with explicit fused multiply-adds (the file is compiled with -ffp-contract=off,
    // which had left a multiply and an add per term: twice the fp64 issue slots of the loop).
    const int x0 = tid * XPER;
    double acc[XPER], win[XPER];
#pragma unroll
    for (int qx = 0; qx < XPER; ++qx) {
        acc[qx] = 0;
        win[qx] = kern2[kswz(qx + W)];              // K(qx - d) at d = -W
    }
    // step st of a group handles d = db + st with the window rotated by st: K(qx - d) sits in win[(qx - st) & 7];
    // afterwards the slot of qx = 7 is refilled with K(0 - (d + 1))
    for (int db = -W; db <= W + XPER - 1; db += XPER) {
        double wk[XPER], nxt[XPER];
#pragma unroll
        for (int st = 0; st < XPER; ++st) {
            const int v = x0 + db + st;
            wk[st] = khist[kswz(min(max(v, -1), SPAN) + 1)];
            nxt[st] = kern2[kswz(abs(db + st + 1))];
        }
#pragma unroll
        for (int st = 0; st < XPER; ++st) {
#pragma unroll
            for (int qx = 0; qx < XPER; ++qx) acc[qx] = __builtin_fma(wk[st], win[(qx - st) & 7], acc[qx]);
            win[(7 - st) & 7] = nxt[st];
        }
    }
    double part = 0;
#pragma unroll
    for (int qx = 0; qx < XPER; ++qx)
        if (x0 + qx < SPAN) part += acc[qx];
    const double tot = block_sum(part, red);
#pragma unroll
    for (int qx = 0; qx < XPER; ++qx)
        if (x0 + qx < SPAN) pdf[x0 + qx] = acc[qx] / tot;
    return 0;
}

__device__ __forceinline__ int kde_block(const int32_t* lens, int n, int* hist, double* kern2, double* khist, double* pdf, double* red, int* flag) {
    kde_clear(hist, flag);
    __syncthreads();
    const KdeLens in = kde_collect(lens, n, hist, red, flag);
    return kde_finish(in, n, hist, kern2, khist, pdf, red, flag);
}

__global__ __launch_bounds__(NT) void pe_kde_kernel(GridArgs a) {
    __shared__ int hist[SPAN];
    __shared__ double kern[KERN2];
    __shared__ double khist[KHIST];
    __shared__ double red[NT / 64];
    __shared__ int flag[4];
    const int g = blockIdx.x;
    const tredgpu_unit_params u = a.units[g];
    const int rc = kde_block(a.global_lens + u.pe_off, u.n_global, hist, kern, khist, a.kde_pdf + (size_t)g * SPAN, red, flag);
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

constexpr int NR = 256;     // threads per workgroup of grid_reduce_kernel

// Where one unit's tables live in the scratch pool (offsets in doubles from the unit's base)
struct SlotLayout {
    int32_t obs, rowoff, far1, far2, near1, near2, rept, roll1, roll2, ml, total;
};
__device__ inline SlotLayout unit_layout(int nrow, int ncol, int nt, bool run_pe, bool haploid, int dmax, int n_near) {
    SlotLayout L;
    int o = 0;
    L.obs = o;    o += (int)((sizeof(Obs) + 7) / 8);
    L.rowoff = o; o += (nrow + 2) / 2;
    L.far1 = o;   o += nrow;
    L.far2 = o;   o += nrow;
    L.near1 = o;  o += nrow * n_near;
    L.near2 = o;  o += nrow * n_near;
    L.rept = o;   o += dmax + 1;
    L.roll1 = o;  o += run_pe ? nrow * ((nt + 31) & ~31) : 0;   // rows padded to 32 entries
    L.roll2 = o;  o += run_pe && !haploid ? ncol * nt : 0;
    L.ml = o;     o += nrow * ncol;
    L.total = (o + 15) & ~15;   // slots start on 128-byte lines
    return L;
}

// Per-unit record handed from grid_prepare_kernel to the other two kernels.
// status: the tredgpu_call status, or UNIT_DEFERRED (no room in the pool this pass), UNIT_SKIP (done in an
// earlier pass)
constexpr int UNIT_DEFERRED = 100;
constexpr int UNIT_SKIP = 101;
struct UnitDesc {
    int32_t status, run_pe, n_rept, n_pairs;
    int32_t period, readlen, ploidy, ref_len, minpe, n_target, tl_off, pad0;
    int32_t t1, t2, mp_eff, h_far;
    int32_t nrow, ncol, nb, hmaxv;
    Axis ax1, ax2;
    int32_t nbn, n_near;   // near columns (h2 < h_far): the first nbn base entries + the first n_near - nbn arithmetic ones
    int32_t cutoff_risk, is_expansion, is_recessive, pad1;
    double half_depth, lgam_rept, logsmall;
    int64_t slot_off;      // doubles from the start of the pool
    SlotLayout lay;
    int32_t item_base, n_items;
};

// Device-side counters of one pass (zeroed by the launch).
// Every counter that many workgroups add to exists TQ times, each copy on its own 64-byte line, and unit g (work item
// t) belongs to copy g % TQ: 30 000 workgroups taking their tickets from ONE address cost the kernels 0.36 ms each --
// device-scope atomics on one address are carried out one after the other, ~12 ns apiece, wherever they come from
// (measured: a kernel that does nothing but take tickets, 0.372 ms for 30 000 units, 0.011 ms without the atomic).
// A workgroup starts at queue blockIdx % TQ and moves on when a queue runs dry (next_ticket): balancing stays dynamic.
// The scratch pool is split the same way into PQ <= TQ sub-pools (bump allocation per sub-pool; PQ = 1 when a single
// unit could need more than a TQ-th of the pool), the work-item list into TQ regions.
constexpr int TQ = 16;
struct CounterLine {
    int32_t v;
    int32_t pad[15];
};
struct CounterLine64 {
    unsigned long long v;
    int32_t pad[14];
};
struct GridCounters {
    CounterLine64 pool_used[TQ];   // doubles handed out, per sub-pool
    CounterLine n_items[TQ];       // work items of the units of queue q (the q-th region of the item list)
    CounterLine next_kde[TQ], next_prepare[TQ], next_reduce[TQ], next_item[TQ];
    CounterLine n_deferred;
};
static_assert(sizeof(CounterLine) == 64 && sizeof(CounterLine64) == 64 && sizeof(GridCounters) == 64 * (6 * TQ + 1), "counter block layout");

// The next unit (or work item) of a kernel's ticket queues, n when all are taken.  Queue q hands out q, q + TQ, ...
// (limit[q] entries when `limit` is given: the work items of a region); called by one thread per workgroup, which
// keeps `q` and `tried` between calls.
struct TicketState {
    int q, tried;
};
__device__ __forceinline__ int next_ticket(CounterLine* queues, TicketState& ts, int n) {
    while (ts.tried < TQ) {
        const int g = atomicAdd(&queues[ts.q].v, 1) * TQ + ts.q;
        if (g < n) return g;
        ts.q = (ts.q + 1) & (TQ - 1);
        ++ts.tried;
    }
    return n;
}
static_assert((TQ & (TQ - 1)) == 0, "queue index wraps by masking");

constexpr int CB = 64;    // columns per work item of grid_pairs_kernel: one lane owns one h2 for all the item's rows
constexpr int RG = 128;   // rows per item
constexpr int TC = 32;    // spanning pairs per pass over the rows (the lane's roll(h2) values stay in registers)
static_assert(TC == 32, "roll1 rows are padded to 32 entries (unit_layout)");

// short grids (up to RS rows): one item takes all the column blocks, so the unit is looked up once
constexpr int RS = 16;
__device__ __forceinline__ int unit_items(int nrow, int ncol) {
    return nrow <= RS ? 1 : ((ncol + CB - 1) / CB) * ((nrow + RG - 1) / RG);
}

// A unit's parameters as wave-uniform values: the loads go through the vector path (the kernels store to
// memory the compiler cannot tell apart), readfirstlane moves every field to a scalar register.
__device__ __forceinline__ tredgpu_unit_params uniform_unit(const tredgpu_unit_params* p) {
    static_assert(sizeof(tredgpu_unit_params) == 16 * sizeof(int), "tredgpu_unit_params is 16 dwords");
    union { tredgpu_unit_params u; int w[16]; } x;
    x.u = *p;
#pragma unroll
    for (int k = 0; k < 16; ++k) x.w[k] = __builtin_amdgcn_readfirstlane(x.w[k]);
    return x.u;
}

// A unit descriptor as wave-uniform values (see uniform_unit): the pairs kernel keeps it live through its loops.
__device__ __forceinline__ UnitDesc uniform_desc(const UnitDesc* p) {
    static_assert(sizeof(UnitDesc) % sizeof(int) == 0, "UnitDesc is a whole number of dwords");
    constexpr int NW = sizeof(UnitDesc) / sizeof(int);
    union { UnitDesc d; int w[NW]; } x;
    x.d = *p;
#pragma unroll
    for (int k = 0; k < NW; ++k) x.w[k] = __builtin_amdgcn_readfirstlane(x.w[k]);
    return x.d;
}

struct PrepShared {
    Obs obs;
    union {
        int hist[SPAN];                    // raw histograms while the lists are built ...
        int row_off[GRID_MAX_ROWS + 1];    // ... then the per-row pair counts / dump offsets
    };
    double step[40];                       // the step-size row of the unit's period
    int tl[SPAN];                          // the spanning pairs' lengths as indices into the rolled pdf
    double pdf[SPAN];                      // the unit's KDE (copied from grid_kde_kernel's output when the term is used)
    long long slot_off;
    int flag, status, unit;
};

__device__ PairCtx make_ctx(const UnitDesc& d, const ModelConst& M, const Obs* obs, const double* pdf, const int32_t* tl) {
    PairCtx C;
    C.M = &M;
    C.step = M.step[d.period <= 6 ? d.period - 1 : 5];  // models.py:54-60
    C.obs = obs;
    C.pdf = pdf;
    C.tl = tl + d.tl_off;
    C.n_target = d.n_target;
    C.period = d.period; C.readlen = d.readlen; C.t1 = d.t1; C.t2 = d.t2; C.mp_eff = d.mp_eff;
    C.ref_len = d.ref_len; C.minpe = d.minpe; C.n_rept = d.n_rept; C.run_pe = d.run_pe != 0;
    C.half_depth = d.half_depth;
    C.lgam_rept = d.lgam_rept;
    C.small = M.small; C.really_small = M.really_small;
    C.logsmall = d.logsmall;
    return C;
}

// ---- kernel 0: the paired-end model of every unit that has one (models.py:131-132, 428-439) ----------------
// The KDE used to sit inside grid_prepare_kernel: its window registers and 17 KB of tables held that kernel at 168
// VGPRs + a spill frame and 5 workgroups per CU although the rest of it waits on memory, not on arithmetic.  Here the
// fp64 convolution has a kernel of its own; the pdf goes to unit_pdf[g][0..1000) and the reference's outcome
// (0, -2 singular, -6 length out of range) to unit_kde_rc[g].  Whether the paired-end term is used at all
// (`run_pe`, models.py:234-236) follows from the unit's histograms alone: the largest FULL / PREF sizes and the
// PREF reads above max_full + period -- the same numbers grid_prepare_kernel derives from its sparse lists (where
// those overflow, the unit fails with -9 and never looks at this result).  Units whose model exists but is not
// used only get the singularity check (the reference builds the KDE regardless, and raises).
__global__ __launch_bounds__(NT, KDE_WAVES) void grid_kde_kernel(GridArgs a, GridCounters* ctr) {
    __shared__ int hist[SPAN];
    __shared__ double kern[KERN2];
    __shared__ double khist[KHIST];
    __shared__ double red[NT / 64];
    __shared__ int flag[4];
    __shared__ int sh[4];
    const int tid = threadIdx.x;
    TicketState ts = {(int)(blockIdx.x & (TQ - 1)), 0};
    while (true) {
        __syncthreads();
        if (tid == 0) { sh[0] = next_ticket(ctr->next_kde, ts, a.n_units); sh[1] = 0; sh[2] = 0; sh[3] = 0; }
        __syncthreads();
        const int g = __builtin_amdgcn_readfirstlane(sh[0]);
        if (g >= a.n_units) break;
        const tredgpu_unit_params u = uniform_unit(a.units + g);
        if (!(u.n_global >= 100 && u.n_target >= 5)) {
            if (tid == 0) a.unit_kde_rc[g] = 0;
            continue;
        }
        // the unit's histograms (is the paired-end term used?) and its pair lengths are fetched together
        kde_clear(hist, flag);
        __syncthreads();
        const int32_t* fc = a.full_cnt + (size_t)g * a.hist_stride;
        const int32_t* pc = a.pref_cnt + (size_t)g * a.hist_stride;
        int hf = 0, hp = 0;
        for (int h = tid; h < a.hist_stride; h += NT) {
            if (fc[h] > 0) hf = h;
            if (pc[h] > 0) hp = h;
        }
        for (int o = 32; o > 0; o >>= 1) { hf = max(hf, __shfl_down(hf, o, 64)); hp = max(hp, __shfl_down(hp, o, 64)); }
        if ((tid & 63) == 0) { atomicMax(&sh[1], hf); atomicMax(&sh[2], hp); }
        const KdeLens in = kde_collect(a.global_lens + u.pe_off, u.n_global, hist, red, flag);   // (barriers inside)
        const int max_full = __builtin_amdgcn_readfirstlane(sh[1]) * u.period;
        const int max_partial = __builtin_amdgcn_readfirstlane(sh[2]) * u.period;
        int above = 0;
        for (int h = tid; h < a.hist_stride; h += NT)
            if (h * u.period > max_full + u.period) above += pc[h];
        for (int o = 32; o > 0; o >>= 1) above += __shfl_down(above, o, 64);
        if ((tid & 63) == 0) atomicAdd(&sh[3], above);
        __syncthreads();
        const bool run_pe = max_partial >= u.readlen - 27 && __builtin_amdgcn_readfirstlane(sh[3]) > 1;
        int rc;
        // model not used: only the singularity check matters (the reference builds the KDE regardless, and raises when
        // scipy does: kde_of_equal_lengths)
        if (!run_pe) rc = in.lo == in.hi ? kde_of_equal_lengths(in.lo, u.n_global, hist, kern, flag) : 0;
        else rc = kde_finish(in, u.n_global, hist, kern, khist, a.unit_pdf + (size_t)g * SPAN, red, flag);
        if (tid == 0) a.unit_kde_rc[g] = rc;
    }
}

// ---- kernel 1: per-unit preparation -------------------------------------------------------------------
__global__ __launch_bounds__(NT, PREP_WAVES) void grid_prepare_kernel(GridArgs a, int pass, UnitDesc* descs, double* pool,
                                                          unsigned long long subpool_doubles, int n_subpools, int rows_cap,
                                                          int cols_cap, int* item_unit, int item_region,
                                                          GridCounters* ctr) {
    __shared__ PrepShared S;
    const int tid = threadIdx.x;
    const ModelConst& M = *a.model;
    TicketState ts = {(int)(blockIdx.x & (TQ - 1)), 0};
    while (true) {
        __syncthreads();
        if (tid == 0) {
            S.unit = next_ticket(ctr->next_prepare, ts, a.n_units);
            S.status = 0;
            if (pass > 0 && S.unit < a.n_units && a.calls[S.unit].status != UNIT_DEFERRED) S.status = UNIT_SKIP;
        }
        __syncthreads();
        const int g = __builtin_amdgcn_readfirstlane(S.unit);
        if (g >= a.n_units) break;
        if (S.status == UNIT_SKIP) {   // settled in an earlier pass
            if (tid == 0) descs[g].status = UNIT_SKIP;
            continue;
        }
        const tredgpu_unit_params u = uniform_unit(a.units + g);
        const int period = u.period, readlen = u.readlen;
        const int t1 = readlen - 9, t2 = readlen - 18, t3 = readlen - 27;  // models.py:114-116

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
        if (staged) {
            // wavefront 0 compacts the FULL sizes, wavefront 1 the PREF/POST sizes (ballot + prefix popcount)
            if (tid == 0) { S.flag = 0; S.status = 0; }
            __syncthreads();
            const int wv = tid >> 6, ln = tid & 63;
            const int32_t* cnt = S.hist + wv * a.hist_stride;
            int* K = wv ? S.obs.partK : S.obs.fullK;
            int* Cn = wv ? S.obs.partC : S.obs.fullC;
            int nn = 0, rept = 0;
            for (int base = 0; base < a.hist_stride; base += 64) {
                const int h = base + ln;
                const int c = h < a.hist_stride ? cnt[h] : 0;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(c > 0);
                const int idx = nn + __builtin_popcountll(mask & ((1ull << ln) - 1ull));
                if (c > 0 && idx < MAXOBS) { K[idx] = h * period; Cn[idx] = c; }
                nn += __builtin_popcountll(mask);
                if (wv == 0 && h < a.hist_stride) rept += S.hist[2 * a.hist_stride + h];
            }
            for (int o = 32; o > 0; o >>= 1) rept += __shfl_down(rept, o, 64);
            if (ln == 0) {
                if (wv) S.obs.nP = min(nn, MAXOBS); else { S.obs.nF = min(nn, MAXOBS); S.flag = rept; }
                if (nn > MAXOBS) S.status = -9;
            }
            __syncthreads();
            if (tid == 0 && (period < 1 || period >= 18)) S.status = -7;  // step_size_by_period KeyError, models.py:157
        } else if (tid == 0) {
            int nF = 0, nP = 0, rept = 0, st = 0;
            const int32_t* fc = a.full_cnt + (size_t)g * a.hist_stride;
            const int32_t* pc = a.pref_cnt + (size_t)g * a.hist_stride;
            const int32_t* rc = a.rept_cnt + (size_t)g * a.hist_stride;
            for (int h = 0; h < a.hist_stride; ++h) {
                if (fc[h] > 0) { if (nF < MAXOBS) { S.obs.fullK[nF] = h * period; S.obs.fullC[nF] = fc[h]; } ++nF; }
                if (pc[h] > 0) { if (nP < MAXOBS) { S.obs.partK[nP] = h * period; S.obs.partC[nP] = pc[h]; } ++nP; }
                rept += rc[h];
            }
            if (nF > MAXOBS || nP > MAXOBS) st = -9;
            if (period < 1 || period >= 18) st = -7;  // step_size_by_period KeyError, models.py:157
            S.obs.nF = min(nF, MAXOBS);
            S.obs.nP = min(nP, MAXOBS);
            S.flag = rept;
            S.status = st;
        }
        __syncthreads();
        // (LDS reads land in VGPRs; these are the same for every lane and steer everything below)
        const int nF = __builtin_amdgcn_readfirstlane(S.obs.nF), nP = __builtin_amdgcn_readfirstlane(S.obs.nP);
        const int n_rept = __builtin_amdgcn_readfirstlane(S.flag);
        int status = __builtin_amdgcn_readfirstlane(S.status);
        const int max_full = __builtin_amdgcn_readfirstlane(nF ? S.obs.fullK[nF - 1] : 0);
        const int max_partial = __builtin_amdgcn_readfirstlane(nP ? S.obs.partK[nP - 1] : 0);
        int reads_above_full = 0;
        for (int i = 0; i < nP; ++i)
            if (S.obs.partK[i] > max_full + period) reads_above_full += S.obs.partC[i];
        reads_above_full = __builtin_amdgcn_readfirstlane(reads_above_full);
        // observation sizes index the 1000-vectors (models.py:198,206): IndexError past the end
        if (status == 0 && (max_full >= SPAN || max_partial >= SPAN)) status = -3;
        __syncthreads();

        // ---- paired-end model (models.py:131-132, 428-439): built by grid_kde_kernel ----
        const bool have_pe = u.n_global >= 100 && u.n_target >= 5;
        const bool run_pe = max_partial >= t3 && reads_above_full > 1 && have_pe;  // :234-236
        if (status == 0 && have_pe) {
            // the reference builds the KDE whenever the model exists; a singular one raises there
            const int rc = __builtin_amdgcn_readfirstlane(a.unit_kde_rc[g]);
            if (rc) status = rc;
        }
        // the spanning pairs' lengths index the rolled pdf (models.py:471-473): IndexError outside [-1000, 1000).
        // Checked by all threads at once, and kept in LDS for the roll tables below: read one by one from global memory
        // by every thread they were a chain of ~n_target memory latencies per unit.
        const bool tl_staged = u.n_target <= SPAN;
        int* const tl_lds = S.tl;
        if (status == 0 && run_pe) {
            // (a gather from LDS per roll-table entry instead of one from global memory: 1.03 -> 0.97 ms per 30 000 units)
            for (int i = tid; i < SPAN; i += NT) S.pdf[i] = a.unit_pdf[(size_t)g * SPAN + i];
            int bad = 0;
            for (int i = tid; i < u.n_target; i += NT) {
                int x = a.target_lens[u.tl_off + i];
                if (x < 0) x += SPAN;
                if (x < 0 || x >= SPAN) bad = 1;
                if (tl_staged) tl_lds[i] = x;
            }
            if (__syncthreads_or(bad)) status = -3;
        }

        // ---- grid axes (models.py:239-257) ----
        {   // sorted(set(FULL keys) | {max_partial}): the ascending, distinct FULL sizes with max_partial slipped in
            // (when there are PREF/POST reads and no FULL size equals it) -- every entry moves at most one place
            int below = 0, same = 0;
            for (int i = 0; i < nF; ++i) { const int v = S.obs.fullK[i]; below += v < max_partial; same += v == max_partial; }
            const bool insert = nP != 0 && same == 0;
            for (int i = tid; i < nF; i += NT) {
                const int v = S.obs.fullK[i];
                S.obs.base[i + (insert && v > max_partial ? 1 : 0)] = v;
            }
            if (tid == 0) {
                if (insert) S.obs.base[below] = max_partial;
                S.obs.nb = nF + (insert ? 1 : 0);
            }
        }
        __syncthreads();
        const int nb = __builtin_amdgcn_readfirstlane(S.obs.nb);
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
        if (status == 0 && (nrow > rows_cap || ncol > cols_cap)) status = -5;
        if (status == 0 && (nrow == 0 || ncol == 0)) status = -8;
        int hmaxv = 0;
        if (status == 0) {   // marginals are indexed by repeat units: every axis value must fit
            hmaxv = axis_value(ax1, S.obs.base, period, nrow - 1);
            if (u.ploidy != 1) hmaxv = max(hmaxv, axis_value(ax2, S.obs.base, period, ncol - 1));
            if (nb > 0) hmaxv = max(hmaxv, S.obs.base[nb - 1]);
            if (hmaxv / period >= MAXM) status = -5;
        }

        UnitDesc d;
        d.status = status; d.run_pe = run_pe; d.n_rept = n_rept; d.n_pairs = 0;
        d.period = period; d.readlen = readlen; d.ploidy = u.ploidy; d.ref_len = u.ref_len; d.minpe = u.minpe;
        d.n_target = u.n_target; d.tl_off = u.tl_off; d.pad0 = 0;
        d.t1 = t1; d.t2 = t2; d.mp_eff = mp_eff;
        d.h_far = max(max(max_full + 19, mp_eff), t1);
        d.nrow = nrow; d.ncol = ncol; d.nb = nb; d.hmaxv = hmaxv;
        d.ax1 = ax1; d.ax2 = ax2;
        d.nbn = 0; d.n_near = 0;
        d.cutoff_risk = u.cutoff_risk; d.is_expansion = u.is_expansion; d.is_recessive = u.is_recessive; d.pad1 = 0;
        d.half_depth = u.half_depth;
        if (n_rept < GRID_LFACT) d.lgam_rept = M.lfact[n_rept];
        else {
            const double x = (double)n_rept + 1, r = 1.0 / x, r2 = r * r;
            d.lgam_rept = (x - .5) * log(x) - x + .91893853320467274178 + r * (1. / 12 - r2 * (1. / 360 - r2 * (1. / 1260)));
        }
        d.logsmall = M.logsmall;
        d.slot_off = 0; d.item_base = 0; d.n_items = 0;
        if (status != 0) {
            if (tid == 0) descs[g] = d;
            continue;
        }

        // ---- everything the pairs kernel adds up comes from tables filled below: the repeat-only term depends on
        //      dsum = max(h1-L,1) + max(h2-L,1) only, the paired-end term on roll(h)[x_t] per axis value, the
        //      spanning + partial terms on the row alone once h2 >= h_far ("far"), on (row, column) for the few
        //      "near" columns below h_far -- a prefix of the base part and a prefix of the arithmetic part of
        //      the h2 axis, both ascending
        const int rect = nrow * ncol;
        const int dmax = 2 * max(hmaxv - readlen, 1);
        const bool haploid = u.ploidy == 1;
        if (!haploid) {
            for (int k = 0; k < ax2.nb; ++k) d.nbn += S.obs.base[k] < d.h_far;
            int nan = 0;
            if (ax2.n > 0 && d.h_far > ax2.start) nan = min(ax2.n, (d.h_far - ax2.start + period - 1) / period);
            d.n_near = d.nbn + nan;
        }

        // ---- room in the pool and a run of work items; a unit that finds the pool full waits for the next pass
        d.lay = unit_layout(nrow, ncol, u.n_target, run_pe, haploid, dmax, d.n_near);
        d.n_items = unit_items(nrow, ncol);
        if (tid == 0) {
            const int sp = g & (n_subpools - 1), q = g & (TQ - 1);
            const unsigned long long off = atomicAdd(&ctr->pool_used[sp].v, (unsigned long long)d.lay.total);
            S.flag = off + d.lay.total <= subpool_doubles;
            S.slot_off = (long long)(sp * subpool_doubles + off);
            if (S.flag) S.status = q * item_region + atomicAdd(&ctr->n_items[q].v, d.n_items);
            else atomicAdd(&ctr->n_deferred.v, 1);
        }
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(S.flag)) {
            if (tid == 0) { d.status = UNIT_DEFERRED; descs[g] = d; }
            continue;
        }
        {
            const long long so = S.slot_off;
            d.slot_off = ((long long)__builtin_amdgcn_readfirstlane((int)(so >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)so);
        }
        d.item_base = __builtin_amdgcn_readfirstlane(S.status);
        const SlotLayout L = d.lay;
        double* slot = pool + d.slot_off;
        for (int k = tid; k < d.n_items; k += NT) item_unit[d.item_base + k] = g;

        // ---- hand the unit's lists to the slot ----
        Obs* gobs = reinterpret_cast<Obs*>(slot + L.obs);
        {
            const int* src = reinterpret_cast<const int*>(&S.obs);
            int* dst = reinterpret_cast<int*>(gobs);
            for (int k = tid; k < (int)(sizeof(Obs) / sizeof(int)); k += NT) dst[k] = src[k];
        }
        // the step-size row of this period in LDS: every spanning / partial term of the row and near tables below looks
        // it up, and from global memory each look-up sat in the terms' chain
        for (int k = tid; k < 37; k += NT) S.step[k] = M.step[d.period <= 6 ? d.period - 1 : 5][k];
        __syncthreads();
        PairCtx C = make_ctx(d, M, &S.obs, S.pdf, a.target_lens);
        C.step = S.step;

        // ---- rows: count of valid h2 per h1 (h1 <= h2), dump offsets; per-row "far" terms ----
        // For h2 >= h_far the spanning and partial terms no longer depend on h2 (S(k|h2) = 0 for every
        // observed size, alpha is pinned, pdf_partial is clipped at max_partial): bit-identical values,
        // evaluated once per row instead of once per pair.
        int* row_off = reinterpret_cast<int*>(slot + L.rowoff);
        // lanes per table entry (eval_reads): as many as keep the workgroup busy, at most 32
        // ONE group size for the row table and the near table: where the reference's terms are the same numbers for
        // two columns of a row (a near column whose terms have already stopped depending on h2, and the far ones) their
        // sums must come out bit-identical here too -- same lanes, same order -- or an exact tie of two pairs'
        // likelihoods (which the reference breaks by enumeration order) would be broken by the last bit
        int G = 1;
        while (G < 32 && nrow * (1 + d.n_near) * (2 * G) <= NT) G *= 2;
        const int gl = tid & (G - 1);
        for (int i = tid / G; i < nrow; i += NT / G) {
            const int h1 = axis_value(ax1, S.obs.base, period, i);
            int cnt = 0;   // columns with h2 >= h1: base entries one by one, the arithmetic part in closed form
            if (u.ploidy == 1) cnt = 1;
            else {
                for (int k = 0; k < ax2.nb; ++k) cnt += S.obs.base[k] >= h1;
                const int m0 = h1 <= ax2.start ? 0 : (h1 - ax2.start + period - 1) / period;
                cnt += max(0, ax2.n - m0);
            }
            double f1, f2;
            eval_reads(C, h1, haploid ? h1 : max(d.h_far, h1), f1, f2, gl, G);   // (one allele: the only column is h2 = h1)
            if (gl == 0) {
                S.row_off[i] = cnt;
                slot[L.far1 + i] = f1;
                slot[L.far2 + i] = f2;
            }
        }
        __syncthreads();
        if (tid < 64) {   // exclusive scan of the row counts: wavefront 0, 64 rows per step
            int carry = 0;
            for (int b0 = 0; b0 < nrow; b0 += 64) {
                const int i = b0 + tid;
                const int c = i < nrow ? S.row_off[i] : 0;
                int incl = c;
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (tid >= o) incl += up;
                }
                if (i < nrow) S.row_off[i] = carry + incl - c;
                carry += __shfl(incl, 63, 64);
            }
            if (tid == 0) S.row_off[nrow] = carry;
        }
        __syncthreads();
        d.n_pairs = __builtin_amdgcn_readfirstlane(S.row_off[nrow]);
        if (a.grid_dump != nullptr)
            for (int i = tid; i <= nrow; i += NT) row_off[i] = S.row_off[i];

        // ---- spanning + partial terms of the near columns
        for (int k = tid / G; k < nrow * d.n_near; k += NT / G) {
            const int i = k / d.n_near, c = k - i * d.n_near;
            const int h1 = axis_value(ax1, S.obs.base, period, i);
            const int h2 = c < d.nbn ? S.obs.base[c] : ax2.start + (c - d.nbn) * period;
            double f1 = 0, f2 = 0;
            if (h1 <= h2) eval_reads(C, h1, h2, f1, f2, gl, G);     // (h1, h2 are the group's: all its lanes or none)
            if (gl == 0) {
                slot[L.near1 + k] = f1;
                slot[L.near2 + k] = f2;
            }
        }
        // ---- the repeat-only table
        {
            double* rept_tab = slot + L.rept;
            if (dmax < rect) {
                // fewer table entries than pairs: fill every entry
                for (int x = 2 + tid; x <= dmax; x += NT) rept_tab[x] = rept_term(C, x);
            } else {
                // only the dsum values that occur: {2} U {1 + d} U {d + d'} for d, d' in D = {h - L > 0}
                for (int x = 2 + tid; x <= dmax; x += NT) rept_tab[x] = 1.0;   // 1.0 = unset (terms are <= 0)
                __syncthreads();
                int i = tid / ncol, j = tid - i * ncol;
                for (int pos = tid; pos < rect; pos += NT) {
                    const int h1 = axis_value(ax1, S.obs.base, period, i);
                    const int h2 = haploid ? h1 : axis_value(ax2, S.obs.base, period, j);
                    if (h1 <= h2) rept_tab[max(h1 - readlen, 1) + max(h2 - readlen, 1)] = 2.0;  // needed
                    j += NT;
                    while (j >= ncol) { j -= ncol; ++i; }
                }
                __syncthreads();
                for (int x = 2 + tid; x <= dmax; x += NT)
                    if (rept_tab[x] == 2.0) rept_tab[x] = rept_term(C, x);
            }
        }
        // ---- the paired-end tables: .5 * roll(h)[x_t] per row and (two alleles) per column
        if (run_pe) {
            // (no index arithmetic by division: rows as 32 targets x 4 rows per sweep of the workgroup -- roll1 rows are
            //  padded to a multiple of 32 targets, the padding holds .5: .5 + .5 = a factor of 1 --, columns one per
            //  thread with the targets in the inner loop, which is also what makes the transposed roll2 rows coalesced)
            const int nt = u.n_target, ntp = (nt + 31) & ~31;
            static_assert(NT % 32 == 0, "row sweep below");
            auto target_x = [&](int t) {                 // spanning pair t as an index into the rolled pdf
                if (tl_staged) return tl_lds[t];
                int x = C.tl[t];
                return x < 0 ? x + SPAN : x;
            };
            for (int tb = 0; tb < ntp; tb += 32) {
                const int t = tb + (tid & 31);
                const int x = t < nt ? target_x(t) : 0;
                for (int ai = tid >> 5; ai < nrow; ai += NT / 32) {
                    const int h = axis_value(ax1, S.obs.base, period, ai);
                    // the tables hold .5 * roll(h)[x]: the pair's factor is the plain sum of two entries (models.py:469)
                    slot[L.roll1 + (size_t)ai * ntp + t] = t < nt ? .5 * roll_at(C.pdf, C.ref_len, C.minpe, h, x, C.small) : .5;
                }
            }
            if (!haploid) {
                for (int ai = tid; ai < ncol; ai += NT) {
                    const int h = axis_value(ax2, S.obs.base, period, ai);
                    for (int t = 0; t < nt; ++t) {
                        const int x = target_x(t);
                        slot[L.roll2 + (size_t)t * ncol + ai] = .5 * roll_at(C.pdf, C.ref_len, C.minpe, h, x, C.small);   // transposed
                    }
                }
            }
        }
        if (tid == 0) descs[g] = d;
    }
}

__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ---- kernel 2: every pair of every unit ------------------------------------------------------------------
// One wavefront per work item = (unit, 64 columns, up to 128 rows); lane = column.  Everything that depends
// on the column only (h2, roll(h2)[x_t] for 32 spanning pairs at a time) sits in registers while the rows
// stream by; what depends on the row only (h1, the far terms, roll(h1)[x_t]) is fetched one row ahead, the
// 32 roll(h1) values with one coalesced load that v_readlane then hands out.
//
// paired-end term (models.py:460-473) from the tables: log of the running product of
// max(.5 * roll(h1)[x] + .5 * roll(h2)[x], SMALL), flushed every 32 factors (every factor is in [e^-10, 1], so
// 32 of them stay above e^-320: one log per 32 pairs instead of one per pair, O(1e-14) from the reference's
// term-by-term sum, far inside the 1e-6 contract); with more than
// 32 spanning pairs the rows are walked once per 32 and the partial sums parked in the ml buffer.
// (the tables hold .5 * roll(h)[x], so a factor is max(entry + entry, SMALL).)
struct RowIn {
    int h1;
    double f1, f2, r1v;
};

__global__ __launch_bounds__(256, GRID_PAIRS_WAVES) void grid_pairs_kernel(GridArgs a, const UnitDesc* descs, double* pool,
                                                         const int* item_unit, Best* item_best, int item_region,
                                                         GridCounters* ctr) {
    const int lane = threadIdx.x & 63;
    int q = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (TQ - 1))), tried = 0;
    const int region_items = lane < TQ ? ctr->n_items[lane].v : 0;   // lane q: the items of region q
    const double small = a.model->small;
    // the row's 32 roll(h1) values, one buffer per wavefront: written by the wave, read back as LDS broadcasts (the
    // same address in every lane) -- an LDS instruction per value instead of two v_readlane on the VALU, which is
    // what the kernel is short of
    __shared__ double rowvals[4][TC];
    static_assert(TC == 32, "rowvals is written by lane & 31");
    double* const myrow = rowvals[threadIdx.x >> 6];
    while (true) {
        // (queue q hands out the items of region q one by one; q and tried are wave-uniform, lane 0 takes the ticket)
        int t = -1;
        while (tried < TQ) {
            int k = 0;
            if (lane == 0) k = atomicAdd(&ctr->next_item[q].v, 1);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k < __builtin_amdgcn_readlane(region_items, q)) { t = q * item_region + k; break; }
            q = (q + 1) & (TQ - 1);
            ++tried;
        }
        if (t < 0) break;
        const int g = __builtin_amdgcn_readfirstlane(item_unit[t]);
        const UnitDesc d = uniform_desc(descs + g);   // by value, in scalar registers across the stores below
        const SlotLayout L = d.lay;
        double* slot = pool + d.slot_off;
        const int nrow = d.nrow, ncol = d.ncol, period = d.period;
        const bool haploid = d.ploidy == 1;
        const Obs* obs = reinterpret_cast<const Obs*>(slot + L.obs);
        double* mlbuf = slot + L.ml;
        int64_t dump_base = -1;
        if (a.grid_dump != nullptr && d.n_pairs <= a.grid_off[g + 1] - a.grid_off[g]) dump_base = a.grid_off[g];
        const int* row_off = reinterpret_cast<const int*>(slot + L.rowoff);
        const int ncb = (ncol + CB - 1) / CB;
        const int k = t - d.item_base;
        const bool short_grid = nrow <= RS;
        const int rg = short_grid ? 0 : k / ncb;
        const int cb_begin = short_grid ? 0 : k - rg * ncb, cb_end = short_grid ? ncb : cb_begin + 1;
        // the rows of a unit in equal groups (257 rows were 128 + 128 + 1: five one-row items per unit), and the column
        // blocks aligned to the END of the axis: the grids that hold 97 % of all pairs are the upper triangles of
        // 255 ... 257 x 255 ... 257 squares, where column j has j + 1 valid rows -- a partial block at the low end of
        // the axis is done after its width in rows, at the high end it kept one or two lanes busy through every row
        // (a quarter of such a unit's row steps).  Neither changes a number: every pair is computed as before.
        const int nrg = (nrow + RG - 1) / RG, rgs = (nrow + nrg - 1) / nrg;
        const int i_begin = rg * rgs, i_end = min(nrow, i_begin + rgs);
        const int pad = ncb * CB - ncol;
        Best mine; mine.ml = 0; mine.h1 = 0; mine.pos = -1;
        for (int cb = cb_begin; cb < cb_end; ++cb) {
        const int j = cb * CB + lane - pad;
        const bool jin = j >= 0;
        const int jc = jin ? j : 0;
        int maxh2 = INT_MAX;                                               // the largest h2 of the block's columns
        if (!haploid) {
            // nothing of the item above the diagonal: the smallest h1 of its rows exceeds the largest h2 of its columns
            // (an axis is its sorted base entries followed by an ascending arithmetic part)
            const int j1 = cb * CB + CB - 1 - pad;                         // the block's last column (>= 0)
            maxh2 = axis_value(d.ax2, obs->base, period, j1);
            if (j1 >= d.ax2.nb && d.ax2.nb > 0 && cb * CB - pad < d.ax2.nb) maxh2 = max(maxh2, obs->base[d.ax2.nb - 1]);
            maxh2 = __builtin_amdgcn_readfirstlane(maxh2);
            int minh1 = axis_value(d.ax1, obs->base, period, i_begin);
            if (i_begin < d.ax1.nb && i_end > d.ax1.nb) minh1 = min(minh1, d.ax1.start);
            if (__builtin_amdgcn_readfirstlane(minh1) > maxh2) continue;
        }
        const int h2col = haploid ? 0 : axis_value(d.ax2, obs->base, period, jc);
        // near column (h2 < h_far): its index in the near tables, else -1 (the row's far terms apply)
        int jn = -1;
        if (!haploid) {
            if (jc < d.ax2.nb) jn = jc < d.nbn ? jc : -1;
            else jn = jc - d.ax2.nb < d.n_near - d.nbn ? d.nbn + jc - d.ax2.nb : -1;
        }
        const bool tab = d.run_pe != 0;
        const int n = d.n_target;
        const int ntp = (n + TC - 1) & ~(TC - 1);
        const int npass = tab ? ntp / TC : 1;
        const double* roll1 = slot + L.roll1;
        for (int pass = 0; pass < npass; ++pass) {
            const bool last = pass == npass - 1;
            const int t0 = pass * TC;
            const int nq = min(TC, n - t0);   // spanning pairs of this pass (wave-uniform)
            double b[TC];
            if (tab && !haploid) {
                // (all 32 loads in flight together: the index is clamped and the padding selected afterwards -- with a
                //  branch per entry the compiler put a full vmcnt(0) wait in front of every load)
                const double* col = slot + L.roll2 + jc;
                const int last_t = n - 1 - t0;
#pragma unroll
                for (int q = 0; q < TC; ++q) b[q] = col[(size_t)(t0 + min(q, last_t)) * ncol];
#pragma unroll
                for (int q = 0; q < TC; ++q) b[q] = q <= last_t ? b[q] : .5;
            }
            auto load_row = [&](int i) {
                RowIn r;
                r.h1 = axis_value(d.ax1, obs->base, period, i);
                r.f1 = slot[L.far1 + i];
                r.f2 = slot[L.far2 + i];
                r.r1v = .5;
                if (tab) r.r1v = roll1[(size_t)i * ntp + t0 + (lane & (TC - 1))];
                return r;
            };
            // software pipeline over the rows: row i+2's scalars and roll(h1) values are requested, row i+1's
            // repeat-only term and near terms are gathered (its h1 arrived an iteration ago), row i is evaluated
            const int d2col = max((haploid ? 0 : h2col) - d.readlen, 1);
            struct Gathered { double m2, n1, n2; };
            auto gather_row = [&](int i, int h1) {
                Gathered v;
                const int d2 = haploid ? max(h1 - d.readlen, 1) : d2col;
                v.m2 = slot[L.rept + max(h1 - d.readlen, 1) + d2];
                v.n1 = v.n2 = 0;
                if (jn >= 0) { v.n1 = slot[L.near1 + i * d.n_near + jn]; v.n2 = slot[L.near2 + i * d.n_near + jn]; }
                return v;
            };
            RowIn cur = load_row(i_begin);
            RowIn nxt = i_begin + 1 < i_end ? load_row(i_begin + 1) : cur;
            Gathered gcur = {0, 0, 0};
            if (last) gcur = gather_row(i_begin, cur.h1);
            for (int i = i_begin; i < i_end; ++i) {
                RowIn nn = nxt;
                if (i + 2 < i_end) nn = load_row(i + 2);
                Gathered gnxt = {0, 0, 0};
                if (last && i + 1 < i_end) gnxt = gather_row(i + 1, nxt.h1);
                const int h1r = __builtin_amdgcn_readfirstlane(cur.h1);
                if (i >= d.ax1.nb && h1r > maxh2) break;      // (the arithmetic part ascends: no later row reaches the diagonal)
                const int h2r = haploid ? h1r : h2col;
                const bool ok = jin && h1r <= h2r;
                if (__builtin_amdgcn_ballot_w64(ok) != 0) {   // else: row entirely below the diagonal here
                    double lp = 0;
                    if (tab) {
                        // (entries past the unit's n_target are .5 + .5 = 1: whole groups of eight of them are skipped,
                        //  which leaves the product bit-identical -- the median unit has 16 spanning pairs, not 32)
                        double prod = 1.0;
                        myrow[lane & (TC - 1)] = cur.r1v;     // (both halves of the wave hold the same 32 values)
#define TRED_PROD8(Q0, OTHER) _Pragma("unroll") for (int q = (Q0); q < (Q0) + 8; ++q) { \
                            const double av = myrow[q]; prod *= fmax(av + (OTHER), small); }
#define TRED_PROD(OTHER) TRED_PROD8(0, OTHER) \
                        if (nq > 8) { TRED_PROD8(8, OTHER) if (nq > 16) { TRED_PROD8(16, OTHER) if (nq > 24) { TRED_PROD8(24, OTHER) } } }
                        if (haploid) { TRED_PROD(av) }   // both alleles are the row's h1: .5 * roll + .5 * roll
                        else { TRED_PROD(b[q]) }
#undef TRED_PROD
#undef TRED_PROD8
                        lp = pos_log(prod);
                    }
                    if (ok) {
                        const int pos = i * ncol + j;
                        double m3 = pass == 0 ? 0.0 : mlbuf[pos];
                        if (tab) m3 += lp;
                        if (!last) mlbuf[pos] = m3;
                        else {
                            const double m0 = jn >= 0 ? gcur.n1 : cur.f1;
                            const double m1 = jn >= 0 ? gcur.n2 : cur.f2;
                            const double m2 = gcur.m2;
                            const double ml = m0 + m1 + m2 + m3;  // models.py:269
                            mlbuf[pos] = ml;
                            Best bb; bb.ml = ml; bb.h1 = h1r; bb.pos = pos;
                            if (better(bb, mine)) mine = bb;
                            if (dump_base >= 0) {
                                int within = 0;  // valid columns before j in this row
                                if (!haploid)
                                    for (int jj = 0; jj < j; ++jj) within += axis_value(d.ax2, obs->base, period, jj) >= h1r;
                                double* o = a.grid_dump + (dump_base + row_off[i] + within) * 6;
                                o[0] = h1r; o[1] = h2r; o[2] = m0; o[3] = m1; o[4] = m2; o[5] = m3;
                            }
                        }
                    }
                }
                cur = nxt; nxt = nn; gcur = gnxt;
            }
        }
        }   // column blocks of the item
        // arg-max of the item with key (ml, -h1), first in enumeration order (models.py:299)
        for (int o = 32; o > 0; o >>= 1) {
            Best other;
            other.ml = __shfl_down(mine.ml, o, 64);
            other.h1 = __shfl_down(mine.h1, o, 64);
            other.pos = __shfl_down(mine.pos, o, 64);
            if (better(other, mine)) mine = other;
        }
        if (lane == 0) item_best[t] = mine;
    }
}

// ---- kernel 3: per-unit reductions --------------------------------------------------------------------------
struct ReduceShared {
    double ph1[MAXM], ph2[MAXM];
    double cum1[GRID_MAX_ROWS];   // row sums first, then the running sum of P_h1
    double cum2[MAXM];
    double colsum[GRID_MAX_COLS];
    double red[NR / 64];
    Best bred[NR / 64];
    int lo[2], brk[2], lastnz[2];
    int jn;      // sparse joint entries found so far
    int unit;
    int base[MAXOBS + 1];   // the unit's observed sizes (base part of the axes)
};

__device__ __forceinline__ double block_sum_r(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < NR / 64; ++w) t += red[w];
    return t;
}

template <bool JOINT>   // JOINT: also the sparse joint distribution (tredgpu_likelihood_grid_joint)
__global__ __launch_bounds__(NR, GRID_REDUCE_WAVES) void grid_reduce_kernel(GridArgs a, const UnitDesc* descs, double* pool,
                                                         const Best* item_best, GridCounters* ctr) {
    __shared__ ReduceShared S;
    const int tid = threadIdx.x;
    TicketState ts = {(int)(blockIdx.x & (TQ - 1)), 0};
    while (true) {
        __syncthreads();
        if (tid == 0) S.unit = next_ticket(ctr->next_reduce, ts, a.n_units);
        __syncthreads();
        const int g = __builtin_amdgcn_readfirstlane(S.unit);
        if (g >= a.n_units) break;
        if (descs[g].status == UNIT_SKIP) continue;   // written in an earlier pass
        const UnitDesc d = descs[g];
        const SlotLayout L = d.lay;
        tredgpu_call call;
        call.status = d.status; call.n_pairs = 0; call.h1 = call.h2 = -1;
        call.ci[0] = call.ci[1] = call.ci[2] = call.ci[3] = 0;
        call.run_pe = d.run_pe; call.pad = 0; call.lik = -1; call.pp = -1;
        if (d.status != 0) {
            if (tid == 0) {
                a.calls[g] = call;
                if (JOINT) { a.joint_n[g] = 0; a.joint_total[g] = 0; }
            }
            if (a.marg != nullptr)
                for (int m = tid; m < 2 * a.marg_stride; m += NR) a.marg[(size_t)g * 2 * a.marg_stride + m] = 0;
            continue;
        }
        double* slot = pool + d.slot_off;
        const double* mlbuf = slot + L.ml;
        const Obs* obs = reinterpret_cast<const Obs*>(slot + L.obs);
        const int nrow = d.nrow, ncol = d.ncol, period = d.period;

        // ---- arg-max over the items' winners ----
        Best mine; mine.ml = 0; mine.h1 = 0; mine.pos = -1;
        for (int k = d.item_base + tid; k < d.item_base + d.n_items; k += NR) {
            const Best b = item_best[k];
            if (better(b, mine)) mine = b;
        }
        for (int o = 32; o > 0; o >>= 1) {
            Best other;
            other.ml = __shfl_down(mine.ml, o, 64);
            other.h1 = __shfl_down(mine.h1, o, 64);
            other.pos = __shfl_down(mine.pos, o, 64);
            if (better(other, mine)) mine = other;
        }
        if ((tid & 63) == 0) S.bred[tid >> 6] = mine;
        const int mlim = min(MAXM, d.hmaxv / period + 1);
        for (int m = tid; m < mlim; m += NR) { S.ph1[m] = 0; S.ph2[m] = 0; }
        if (tid < 2) { S.lo[tid] = MAXM; S.brk[tid] = MAXM; S.lastnz[tid] = 0; }
        if (tid == 0) S.jn = 0;
        for (int k = tid; k < d.nb; k += NR) S.base[k] = obs->base[k];   // axis values in LDS: every row of the sweep asks
        __syncthreads();
        Best top = S.bred[0];
        for (int w = 1; w < NR / 64; ++w) if (better(S.bred[w], top)) top = S.bred[w];
        const double max_ml = top.ml;

        // ---- one pass over the grid: exp(ml - max) (models.py:280-285), the PP sums (:342-368), row sums
        //      (one wave per row, fixed shuffle tree) and per-wave column sums (lane = column, kept in
        //      registers; the four waves' partial sums are added up in wave order below).  Pairs below the
        //      diagonal were never written and count as 0.
        const int lane = tid & 63, wv = tid >> 6;
        const int nk = (ncol + 63) >> 6;
        double all = 0, path = 0, uniq = 0;
        constexpr bool want_joint = JOINT;
        const double small = a.model->small;
        const int jcap = want_joint ? (int)min((long long)(a.joint_off[g + 1] - a.joint_off[g]), 0x7fffffffLL) : 0;
        // KC = column chunks a lane owns per sweep (4 for grids up to 256 columns wide, else 8; wider grids
        // take a second sweep over the rows for chunks 8..15), k0 = first chunk of the sweep
        auto sweep = [&](auto kc_tag, const int k0) {
            constexpr int KC = decltype(kc_tag)::value;
            int h2k[KC];
            unsigned colpath = 0;   // bit k: the column's h2 decides "pathological" and says yes
            const bool by_col = d.ploidy != 1 && ((d.is_expansion != 0) != (d.is_recessive != 0));   // hi decides
    #pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int j = lane + 64 * (k0 + k);
                h2k[k] = k0 + k < nk && j < ncol ? axis_value(d.ax2, S.base, period, j) : -1;   // -1: no column
                const int hi = h2k[k] / period;
                if (by_col && h2k[k] >= 0 && (d.is_expansion ? hi >= d.cutoff_risk : hi <= d.cutoff_risk)) colpath |= 1u << k;
            }
            // sparse joint distribution: the extended axes can list a value twice (models.py:251-252); the pairs of
            // the second occurrence repeat those of the first and the reference's dict keeps one of them
            unsigned coldup = 0;
            if (want_joint && d.ploidy != 1) {
    #pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int j = lane + 64 * (k0 + k);
                    if (h2k[k] >= 0 && j >= d.ax2.nb)
                        for (int q = 0; q < d.ax2.nb; ++q) if (S.base[q] == h2k[k]) coldup |= 1u << k;
                }
            }
            double colacc[KC];
    #pragma unroll
            for (int k = 0; k < KC; ++k) colacc[k] = 0;
            // two rows of the wave per step (i and i + 4): two independent chains of loads, exps and the row-sum
            // reduction in one straight line of code -- the kernel's time per unit is this chain's latency
            auto one_row = [&](const int i) -> double {
                const int h1 = __builtin_amdgcn_readfirstlane(axis_value(d.ax1, S.base, period, i));
                const int lo = h1 / period;
                bool rowdup = false;
                if (want_joint && i >= d.ax1.nb)
                    for (int q = 0; q < d.ax1.nb; ++q) rowdup |= S.base[q] == h1;
                // ploidy 1: h2 = h1, one column; both alleles equal, so lo decides whatever the inheritance
                const bool rowpath = !by_col && (d.is_expansion ? lo >= d.cutoff_risk : lo <= d.cutoff_risk);
                double v[KC];
    #pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const bool ok = k0 + k < nk && (d.ploidy == 1 ? lane == 0 && k == 0 : h2k[k] >= h1);
                    v[k] = ok ? mlbuf[i * ncol + lane + 64 * (k0 + k)] : 0.0;
                }
                double acc = 0;
    #pragma unroll
                for (int k = 0; k < KC; ++k) {
                    if (k0 + k >= nk) break;
                    const bool ok = d.ploidy == 1 ? lane == 0 && k == 0 : h2k[k] >= h1;
                    const double e = ok ? exp(v[k] - max_ml) : 0.0;
                    acc += e;
                    colacc[k] += e;
                    if (rowpath || ((colpath >> k) & 1u)) path += e;
                    if (want_joint && ok && !rowdup && !((coldup >> k) & 1u)) {
                        uniq += e;
                        if (e >= small) {
                            const int at = atomicAdd(&S.jn, 1);
                            if (at < jcap) {
                                double* o = a.joint + (a.joint_off[g] + at) * 3;
                                o[0] = h1;
                                o[1] = d.ploidy == 1 ? h1 : h2k[k];
                                o[2] = e;
                            }
                        }
                    }
                }
                return acc;
            };
            for (int i = wv; i < nrow; i += 2 * (NR / 64)) {
                const int i2 = i + NR / 64;
                const bool two = i2 < nrow;               // wave-uniform
                double acc0 = one_row(i), acc1 = 0;
                if (two) acc1 = one_row(i2);
                all += acc0;
                all += acc1;
                acc0 = wave_sum_to_last(acc0);
                acc1 = wave_sum_to_last(acc1);
                if (lane == 63) {
                    S.cum1[i] = k0 == 0 ? acc0 : S.cum1[i] + acc0;
                    if (two) S.cum1[i2] = k0 == 0 ? acc1 : S.cum1[i2] + acc1;
                }
            }
            // column sums: wave 0 stores, waves 1..3 add in turn
            for (int w = 0; w < NR / 64; ++w) {
                if (wv == w) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int j = lane + 64 * (k0 + k);
                        if (k0 + k < nk && j < ncol) S.colsum[j] = w == 0 ? colacc[k] : S.colsum[j] + colacc[k];
                    }
                }
                __syncthreads();
            }
        };
        if (nk <= 4) sweep(std::integral_constant<int, 4>(), 0);
        else {
            sweep(std::integral_constant<int, 8>(), 0);
            if (nk > 8) sweep(std::integral_constant<int, 8>(), 8);
        }
        // marginal P_h2 by distinct h2 value.  The extended axis can list a value twice (base part +
        // arithmetic part, models.py:251-252): the first occurrence owns the sum of both columns.
        if (d.ploidy != 1) {
            for (int j = tid; j < ncol; j += NR) {
                const int h2 = axis_value(d.ax2, S.base, period, j);
                int twin = -1;
                if (j < d.ax2.nb) {
                    const int dd = h2 - d.ax2.start;
                    if (d.ax2.n > 0 && dd >= 0 && dd % period == 0 && dd / period < d.ax2.n) twin = d.ax2.nb + dd / period;
                } else {
                    bool dup = false;
                    for (int k = 0; k < d.ax2.nb; ++k) dup |= S.base[k] == h2;
                    if (dup) continue;  // owned by the base occurrence
                }
                const int m = h2 / period;
                if (m < MAXM) S.ph2[m] = twin >= 0 ? S.colsum[j] + S.colsum[twin] : S.colsum[j];
            }
        }
        all = block_sum_r(all, S.red);
        path = block_sum_r(path, S.red);   // (its barriers also publish the row sums and S.ph2)
        if (want_joint) {
            uniq = block_sum_r(uniq, S.red);
            if (tid == 0) { a.joint_n[g] = S.jn; a.joint_total[g] = uniq; }
        }
        // marginal P_h1 by distinct h1 value, rows merged by key in row order: like the columns, a value can be listed
        // twice (once in the base part, once in the arithmetic part) and the base occurrence owns base + twin -- the
        // order in which a serial walk over the rows adds them up
        for (int i = tid; i < nrow; i += NR) {
            const int h1 = axis_value(d.ax1, S.base, period, i);
            int twin = -1;
            if (i < d.ax1.nb) {
                const int dd = h1 - d.ax1.start;
                if (d.ax1.n > 0 && dd >= 0 && dd % period == 0 && dd / period < d.ax1.n) twin = d.ax1.nb + dd / period;
            } else {
                bool dup = false;
                for (int k = 0; k < d.ax1.nb; ++k) dup |= S.base[k] == h1;
                if (dup) continue;  // owned by the base occurrence
            }
            const int m = h1 / period;
            if (m < MAXM) S.ph1[m] = twin >= 0 ? S.cum1[i] + S.cum1[twin] : S.cum1[i];
        }
        __syncthreads();
        if (d.ploidy == 1) {
            for (int m = tid; m < mlim; m += NR) S.ph2[m] = S.ph1[m];  // h2 == h1 for every pair
            __syncthreads();
        }
        // ---- calc_CI, models.py:319-340 on each marginal: running sums by one lane each, then every
        //      thread tests its own bins against the 2.5 % / 97.5 % marks ----
        if (tid == 0 || tid == 64) {
            // (eight bins per step: the loads of a step do not wait for one another, only the additions form a chain --
            //  bin by bin the loop ran at one LDS round trip per bin, a third of the kernel's time per unit)
            const double* P = tid ? S.ph2 : S.ph1;
            double* cum = tid ? S.cum2 : S.cum1;
            double c = 0;
            int m = 0;
            for (; m + 8 <= mlim; m += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = P[m + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) { c += v[q]; v[q] = c; }
#pragma unroll
                for (int q = 0; q < 8; ++q) cum[m + q] = v[q];
            }
            for (; m < mlim; ++m) { c += P[m]; cum[m] = c; }
        }
        __syncthreads();
        for (int which = 0; which < 2; ++which) {
            const double* P = which ? S.ph2 : S.ph1;
            const double* cum = which ? S.cum2 : S.cum1;
            const double total = cum[mlim - 1];
            int last = 0, lo = MAXM, brk = MAXM;      // (one LDS atomic per wave and mark, not one per bin)
            for (int m = tid; m < mlim; m += NR) {
                if (P[m] == 0) continue;
                last = max(last, m);
                if (cum[m] > .025 * total) lo = min(lo, m);
                if (cum[m] > .975 * total) brk = min(brk, m);
            }
            last = wave_minmax_to_last<true>(last);
            lo = wave_minmax_to_last<false>(lo);
            brk = wave_minmax_to_last<false>(brk);
            if ((tid & 63) == 63) {
                atomicMax(&S.lastnz[which], last);
                atomicMin(&S.lo[which], lo);
                atomicMin(&S.brk[which], brk);
            }
        }
        __syncthreads();
        if (tid == 0) {
            for (int which = 0; which < 2; ++which) {
                call.ci[2 * which] = S.lo[which] < MAXM ? S.lo[which] : 0;
                call.ci[2 * which + 1] = S.brk[which] < MAXM ? S.brk[which] : S.lastnz[which];
            }
            const int bi = top.pos / ncol, bj = top.pos - bi * ncol;
            call.h1 = axis_value(d.ax1, S.base, period, bi);
            call.h2 = d.ploidy == 1 ? call.h1 : axis_value(d.ax2, S.base, period, bj);
            call.lik = max_ml;
            const double pp = path / all;
            call.pp = pp < 1 ? pp : 1;
            call.n_pairs = d.n_pairs;
            call.status = 0;
            if (a.grid_dump != nullptr && d.n_pairs > a.grid_off[g + 1] - a.grid_off[g]) call.status = -4;
            a.calls[g] = call;
        }
        if (a.marg != nullptr) {
            for (int m = tid; m < a.marg_stride; m += NR) {
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

namespace {
__global__ void unit_max_kernel(const tredgpu_unit_params* units, int n, int* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int v = g < n ? units[g].maxinsert : 0, w = g < n ? units[g].n_target : 0;
    for (int o = 32; o > 0; o >>= 1) { v = max(v, __shfl_down(v, o, 64)); w = max(w, __shfl_down(w, o, 64)); }
    if ((threadIdx.x & 63) == 0) {
        if (v > 0) atomicMax(out, v);
        if (w > 0) atomicMax(out + 1, w);
    }
}
}  // namespace

hipError_t launch_unit_max(const tredgpu_unit_params* units, int n_units, int* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(int), s);
    if (e != hipSuccess || n_units <= 0) return e;
    unit_max_kernel<<<(n_units + 255) / 256, 256, 0, s>>>(units, n_units, out);
    return hipGetLastError();
}

size_t grid_desc_bytes() { return sizeof(UnitDesc); }

size_t grid_counter_bytes() { return sizeof(GridCounters); }

size_t grid_items_cap(int rows_cap, int cols_cap) {   // most work items one unit can make
    return (size_t)((cols_cap + CB - 1) / CB) * ((rows_cap + RG - 1) / RG);
}

size_t grid_item_bytes() { return sizeof(int) + sizeof(Best); }

// largest slot a unit within the caps can ask for (nt_max = most spanning pairs of any unit)
size_t grid_slot_doubles_max(int rows_cap, int cols_cap, int nt_max) {
    const size_t ntp = ((size_t)std::max(nt_max, 0) + 31) & ~(size_t)31;
    return (sizeof(Obs) + 7) / 8 + (size_t)rows_cap * 3 + 2 + 2 * (size_t)18 * MAXM + 2 + (size_t)rows_cap * ntp +
           (size_t)cols_cap * (size_t)std::max(nt_max, 0) + 3 * (size_t)rows_cap * cols_cap + 16;
}

// One pass over all units: prepare (takes pool room per unit) -> pairs -> reduce, all on stream s.  Units that
// found the pool full are marked UNIT_DEFERRED in calls[].status and counted in the counter block's
// n_deferred; the caller runs further passes (pass > 0 touches only those) until none is left.
// items: item_cap ints (item -> unit), then item_cap arg-max records; item_cap = grid_item_slots(n_units, ...): TQ
// regions.  pool_doubles is used as grid_subpools() sub-pools of equal size.
hipError_t launch_grid_pass(const GridArgs& a, int pass, void* descs, double* pool, size_t pool_doubles, int rows_cap,
                            int cols_cap, void* items, size_t item_cap, void* counters, hipStream_t s, int phases) {
    // phases: bit 0 prepare (with the counter reset), bit 1 pairs, bit 2 reduce, bit 3 the paired-end KDEs (first, and
    // in pass 0 only: deferred units keep theirs) -- the caller may launch them one by one to time each (capi.hip
    // brackets them with HIP events)
    if (a.n_units <= 0) return hipSuccess;
    UnitDesc* d = (UnitDesc*)descs;
    GridCounters* ctr = (GridCounters*)counters;
    int* item_unit = (int*)items;
    Best* item_best = (Best*)(item_unit + ((item_cap + 3) & ~(size_t)3));
    const int item_region = (int)(item_cap / TQ);
    const int n_sub = grid_subpools(pool_doubles, rows_cap, cols_cap, a.max_target);
    const unsigned long long sub_doubles = (pool_doubles / n_sub) & ~(unsigned long long)15;   // slots start on 128-byte lines
    if ((phases & 8) && pass == 0) {
        hipError_t e = hipMemsetAsync(counters, 0, sizeof(GridCounters), s);
        if (e != hipSuccess) return e;
        const int kb = a.n_units < KDE_BLOCKS ? a.n_units : KDE_BLOCKS;
        grid_kde_kernel<<<kb, NT, 0, s>>>(a, ctr);
    }
    if (phases & 1) {
        hipError_t e = hipMemsetAsync(counters, 0, sizeof(GridCounters), s);
        if (e != hipSuccess) return e;
        const int pb = a.n_units < PREP_BLOCKS ? a.n_units : PREP_BLOCKS;
        grid_prepare_kernel<<<pb, NT, 0, s>>>(a, pass, d, pool, sub_doubles, n_sub, rows_cap, cols_cap, item_unit,
                                              item_region, ctr);
    }
    if (phases & 2) grid_pairs_kernel<<<2048, 256, 0, s>>>(a, d, pool, item_unit, item_best, item_region, ctr);
    if (phases & 4) {
        const int rb = a.n_units < 2048 ? a.n_units : 2048;
        if (a.joint != nullptr) grid_reduce_kernel<true><<<rb, NR, 0, s>>>(a, d, pool, item_best, ctr);
        else grid_reduce_kernel<false><<<rb, NR, 0, s>>>(a, d, pool, item_best, ctr);
    }
    return hipGetLastError();
}

int grid_deferred_offset() { return (int)offsetof(GridCounters, n_deferred); }

// Sub-pools the scratch pool is used as: as many (a power of two, at most TQ) as still leave room for the largest slot
// in each -- a unit is never too big for its sub-pool, so every pass settles at least one unit per sub-pool.
int grid_subpools(size_t pool_doubles, int rows_cap, int cols_cap, int nt_max) {
    const size_t slot = grid_slot_doubles_max(rows_cap, cols_cap, nt_max) + 16;
    int n = TQ;
    while (n > 1 && pool_doubles / n < slot) n >>= 1;
    return n;
}

// Entries of the work-item list for n_units units: TQ regions, each with room for the units of its queue
size_t grid_item_slots(int n_units, int rows_cap, int cols_cap) {
    return (size_t)TQ * (((size_t)n_units + TQ - 1) / TQ) * grid_items_cap(rows_cap, cols_cap);
}

// Units a sub-pool can be asked to hold at most (the units of one residue class mod the sub-pool count)
size_t grid_units_per_subpool(int n_units, int n_sub) { return ((size_t)n_units + n_sub - 1) / n_sub; }

}  // namespace tredgpu
