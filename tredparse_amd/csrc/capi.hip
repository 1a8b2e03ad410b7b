// capi.hip -- the extern "C" surface of libtredgpu.so (declared in include/tredgpu.h).
// Host-side glue only: argument checks, ladder tables, staging of host buffers, kernel launches.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <unistd.h>

#include "tredgpu_internal.h"
#include "inflater_internal.h"

using namespace tredgpu;

namespace {

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
};

std::string g_create_error;

}  // namespace

struct tredgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // ladders
    std::vector<LadderDesc> h_ladders;
    Buf d_ladders, d_seq;
    int max_templates = 0;  // max over ladders of 2*max_units (or 1)
    int max_ladder_units = 0;
    // model
    Buf d_model;
    bool have_model = false;
    // workspaces (grow-only, reused across calls)
    Buf ws_quads, ws_counter, ws_drop, ws_grid, ws_stats, ws_perm, ws_class, ws_gdesc, ws_gtile, ws_gctr, ws_ucnt, ws_bins, ws_kde;
    int* h_pin = nullptr;  // pinned word for small read-backs
    hipEvent_t sync_ev = nullptr;  // stream_sync's event
    size_t grid_pool_bytes = GRID_POOL_BYTES;   // TREDGPU_GRID_POOL_MB overrides (tuning / tests of the multi-pass path)
    Buf st[40];  // staging for HOST-memory calls
    // pinned arena of the HOST-memory calls: copies from / to the caller's pageable arrays go through it, so that they are
    // truly asynchronous (a hipMemcpyAsync on pageable memory is staged and waited for by the runtime, one by one -- with
    // several driver processes on the device each of those waits queues behind the others' work: 25 per batch)
    struct PinBlock { uint8_t* p; size_t cap, used; };
    std::vector<PinBlock> pin;
    struct PinOut { void* host; const void* pinned; size_t bytes; };
    std::vector<PinOut> pin_out;       // read-backs to hand over at the next stream_sync
    // intermediates of the fused path
    Buf ws_tag, ws_h, ws_score;
    // HIP-event timing of the three main kernels
    struct Timer {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;  // reusable event pairs
        size_t used = 0;
        int64_t launches = 0;
        double total_ms = 0;
    } timers[7];
};

namespace {

int fail(tredgpu_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail((c), -10, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// resolve the recorded event pairs of one timer into (launches, total_ms); stream must be idle
void timer_flush(tredgpu_ctx::Timer& t) {
    for (size_t i = 0; i < t.used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.pool[i].first, t.pool[i].second) == hipSuccess) {
            t.total_ms += ms;
            t.launches += 1;
        }
    }
    t.used = 0;
}

struct ScopedTimer {
    tredgpu_ctx* c;
    tredgpu_ctx::Timer& t;
    hipEvent_t stop = nullptr;
    ScopedTimer(tredgpu_ctx* c_, int which) : c(c_), t(c_->timers[which]) {
        if (t.used >= 256) {  // bounded pool: fold what is already finished
            (void)hipStreamSynchronize(c->stream);
            timer_flush(t);
        }
        if (t.used == t.pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            t.pool.emplace_back(a, b);
        }
        (void)hipEventRecord(t.pool[t.used].first, c->stream);
        stop = t.pool[t.used].second;
    }
    ~ScopedTimer() {
        if (stop) {
            (void)hipEventRecord(stop, c->stream);
            t.used += 1;
        }
    }
};

// `bytes` of pinned memory, valid until the next stream_sync (blocks are kept and reused; a call that needs more than the
// arena holds gets a further block)
void* pin_alloc(tredgpu_ctx* c, size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (auto& b : c->pin)
        if (b.cap - b.used >= bytes) { void* at = b.p + b.used; b.used += bytes; return at; }
    size_t cap = std::max<size_t>(bytes, (size_t)8 << 20);
    if (!c->pin.empty()) cap = std::max(cap, c->pin.back().cap * 2);
    uint8_t* p = nullptr;
    if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    c->pin.push_back({p, cap, bytes});
    return p;
}

// wait for the context's stream; then hand the read-backs that went through the pinned arena to their arrays and start
// the arena afresh (nothing of it is in flight any more)
hipError_t stream_sync(tredgpu_ctx* c) {
    // hipStreamSynchronize spins: a driver process whose genotyping call shares the device with three processes' decoders
    // sits in it for 40 ms per call -- a core per driver burnt on a box that is short of cores (8 CPUs: 36 k genotypes/s).
    // So: a short spin for the calls that are nearly done (the kernel-path benchmark's step ends within microseconds of
    // its last launch), then the event is polled asleep, as the inflater's calls are (inflater_api.hip wait_asleep).
    hipError_t e = hipSuccess;
    if (c->sync_ev == nullptr && hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming) != hipSuccess) c->sync_ev = nullptr;
    if (c->sync_ev != nullptr && (e = hipEventRecord(c->sync_ev, c->stream)) == hipSuccess) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            const hipError_t q = hipEventQuery(c->sync_ev);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { e = q; break; }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) > 300000LL) usleep(100);
        }
    } else
        e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        for (const auto& o : c->pin_out) memcpy(o.host, o.pinned, o.bytes);
    c->pin_out.clear();
    for (auto& b : c->pin) b.used = 0;
    return e;
}

int ensure(tredgpu_ctx* c, Buf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return 0;
    if (b.p) {
        // the buffer may still be in use by enqueued work
        HIPCHK(c, stream_sync(c));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

void release(Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int base_code(char ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

void encode(const char* s, std::vector<int8_t>& out) {
    for (; *s; ++s) out.push_back((int8_t)base_code(*s));
}

// reverse complement on codes (N stays N): bam_parser.py:448-450
std::vector<int8_t> revcomp(const std::vector<int8_t>& v) {
    std::vector<int8_t> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        int c = v[v.size() - 1 - i];
        o[i] = (int8_t)(c == 4 ? 4 : 3 - c);
    }
    return o;
}

int check_sw_range(tredgpu_ctx* c, const tredgpu_sw_params* p, int max_len = 0);

int check_sw_params(tredgpu_ctx* c, const tredgpu_sw_params* p) {
    if (!p) return fail(c, -2, "params is NULL");
    if (p->match < 1 || p->match > 8 || p->mismatch < 0 || p->mismatch > 16 || p->gap_open < 1 ||
        p->gap_open > 16 || p->gap_extend < 1 || p->gap_extend > 16 || p->gap_extend > p->gap_open)
        return fail(c, -2, "scoring out of the supported range (match 1..8, mismatch 0..16, "
                           "1 <= gap_extend <= gap_open <= 16)");
    if (p->flank < 0 || p->flank > 255) return fail(c, -2, "flank out of range");
    return check_sw_range(c, p);
}

int rows_for(int max_len) {
    if (max_len <= 64) return 4;
    if (max_len <= 112) return 7;
    if (max_len <= 160) return 10;
    if (max_len <= 256) return 16;
    if (max_len <= 320) return 20;
    return 32;                          // 16 lanes x 32 rows: the nine row bits of the packed values
}

// Every DP value is (score + (row + col) * gap_extend) << 18 | payload in an int32: the scaled score must stay
// below 2^13 on the longest template (511 columns) for every row the kernel instantiation holds (16 lanes x R rows
// for reads up to params.max_read_len, TREDGPU_MAX_READ_LEN when that is 0).  The default 1/5/7/2 scoring needs
// 1 790 of the 8 192.  max_len > 0: the bound the call resolved (the longest read a HOST-memory call has seen: it may
// select a larger instantiation than the one the arguments alone were checked for).
int check_sw_range(tredgpu_ctx* c, const tredgpu_sw_params* p, int max_len) {
    const int named = max_len > 0 ? max_len : p->max_read_len;
    const int L = named > 0 ? std::min(named, TREDGPU_MAX_READ_LEN) : TREDGPU_ASSUMED_READ_LEN;
    const int need = (16 * rows_for(L) + 511) * p->gap_extend + L * p->match;
    if (need >= 8192)
        return fail(c, -2, "scoring too large for the packed DP values: (rows + 511) * gap_extend + max_read_len * "
                           "match must stay below 8192 (is %d)", need);
    return 0;
}

// copy a host array to a staging buffer; returns device pointer through out
template <typename T>
int stage_in(tredgpu_ctx* c, Buf& b, const T* host, size_t n, const T** out) {
    int rc = ensure(c, b, n * sizeof(T));
    if (rc) return rc;
    if (n) {
        void* h = pin_alloc(c, n * sizeof(T));
        if (!h) return fail(c, -10, "no pinned memory for a %zu-byte copy", n * sizeof(T));
        memcpy(h, host, n * sizeof(T));
        HIPCHK(c, hipMemcpyAsync(b.p, h, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    }
    *out = (const T*)b.p;
    return 0;
}

template <typename T>
int stage_out(tredgpu_ctx* c, Buf& b, size_t n, T** out) {
    int rc = ensure(c, b, n * sizeof(T));
    if (rc) return rc;
    *out = (T*)b.p;
    return 0;
}

template <typename T>
int copy_back(tredgpu_ctx* c, T* host, const T* dev, size_t n) {
    if (n && host) {
        void* h = pin_alloc(c, n * sizeof(T));
        if (!h) return fail(c, -10, "no pinned memory for a %zu-byte copy", n * sizeof(T));
        HIPCHK(c, hipMemcpyAsync(h, dev, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        c->pin_out.push_back({host, h, n * sizeof(T)});       // handed over by stream_sync
    }
    return 0;
}

}  // namespace

extern "C" {

#ifndef TREDGPU_SRC_HASH
#define TREDGPU_SRC_HASH "unknown"
#endif
const char* tredgpu_version(void) { return "tredgpu 0.6 (gfx950) src " TREDGPU_SRC_HASH; }

int tredgpu_create(int device_id, tredgpu_ctx** out) {
    if (!out) return fail(nullptr, -2, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, -3, "no HIP device available (%s); libtredgpu has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, -2, "device %d out of range [0,%d)", device_id, n);
    tredgpu_ctx* c = new tredgpu_ctx();
    c->device = device_id;
    // TREDGPU_CTX_PRIORITY=high: the context's stream above the inflaters' (which are created at the lowest priority): a
    // genotyping call is a dozen short launches, and among other processes' 4 ms decode launches each of them queued
    int lo_prio = 0, hi_prio = 0;
    const char* want = getenv("TREDGPU_CTX_PRIORITY");
    const bool high = want && strcmp(want, "high") == 0;
    const char* split = getenv("TREDGPU_CTX_CUS");
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = (split && atoi(split) > 0) ? tredgpu_front::create_partitioned_stream(&c->stream, hipStreamDefault, 0, 1, device_id)
             : high && hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio) == hipSuccess
                 ? hipStreamCreateWithPriority(&c->stream, hipStreamDefault, hi_prio) : hipStreamCreate(&c->stream)) != hipSuccess) {
        delete c;
        return fail(nullptr, -10, "cannot initialise device %d: %s", device_id, hipGetErrorString(e));
    }
    if (const char* mb = getenv("TREDGPU_GRID_POOL_MB")) {
        const long v = atol(mb);
        if (v > 0) c->grid_pool_bytes = (size_t)v << 20;
    }
    *out = c;
    return 0;
}

void tredgpu_destroy(tredgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)stream_sync(c);
    for (Buf* b : {&c->d_ladders, &c->d_seq, &c->d_model, &c->ws_quads, &c->ws_counter, &c->ws_drop,
                   &c->ws_grid, &c->ws_stats, &c->ws_perm, &c->ws_class, &c->ws_gdesc, &c->ws_gtile, &c->ws_gctr, &c->ws_ucnt, &c->ws_bins, &c->ws_kde, &c->ws_tag, &c->ws_h, &c->ws_score})
        release(*b);
    for (Buf& b : c->st) release(b);
    for (auto& t : c->timers)
        for (auto& ev : t.pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    for (auto& b : c->pin) (void)hipHostFree(b.p);
    c->pin.clear();
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* tredgpu_last_error(const tredgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int tredgpu_sync(tredgpu_ctx* c) {
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_sync(c));
    return 0;
}

void* tredgpu_get_stream(tredgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

int tredgpu_reset_timing(tredgpu_ctx* c) {
    if (!c) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_sync(c));
    for (auto& t : c->timers) { t.used = 0; t.launches = 0; t.total_ms = 0; }
    if (c->ws_stats.p) HIPCHK(c, hipMemsetAsync(c->ws_stats.p, 0, SW_STAT_SLOTS * 8 * sizeof(unsigned long long), c->stream));
    return 0;
}

int tredgpu_get_sw_counters(tredgpu_ctx* c, uint64_t out[8]) {
    if (!c || !out) return -2;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (c->ws_stats.p) {
        // the kernel spreads its per-wave updates over SW_STAT_SLOTS cache lines (one line of 8 counters each)
        std::vector<uint64_t> slots((size_t)SW_STAT_SLOTS * 8);
        HIPCHK(c, hipMemcpyAsync(slots.data(), c->ws_stats.p, slots.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, stream_sync(c));
        for (int k = 0; k < SW_STAT_SLOTS; ++k)
            for (int i = 0; i < 8; ++i) out[i] += slots[(size_t)k * 8 + i];
    }
    HIPCHK(c, stream_sync(c));
    return 0;
}

int tredgpu_get_timing(tredgpu_ctx* c, int which, int64_t* launches, double* total_ms) {
    if (!c) return -2;
    if (which < 0 || which > 6) return fail(c, -2, "which must be one of TREDGPU_KERNEL_*");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, stream_sync(c));
    timer_flush(c->timers[which]);
    if (launches) *launches = c->timers[which].launches;
    if (total_ms) *total_ms = c->timers[which].total_ms;
    return 0;
}

int tredgpu_set_ladders(tredgpu_ctx* c, int32_t n, const char* const* prefix, const char* const* repeat,
                        const char* const* suffix, const int32_t* max_units) {
    if (!c) return -2;
    if (n < 0 || (n > 0 && (!prefix || !repeat || !suffix || !max_units))) return fail(c, -2, "bad ladder arguments");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint32_t> seq;  // letters, 8 per word; every segment starts on a word boundary
    auto append = [&seq](const std::vector<int8_t>& v) {
        const int off = (int)seq.size();
        seq.resize(seq.size() + (v.size() + 7) / 8 + 1, 0x44444444u);
        for (size_t i = 0; i < v.size(); ++i) {
            uint32_t& w = seq[off + i / 8];
            w = (w & ~(0xFu << ((i % 8) * 4))) | ((uint32_t)v[i] << ((i % 8) * 4));
        }
        return off;
    };
    std::vector<LadderDesc> lad((size_t)n);
    int max_t = 1, max_u = 0;
    for (int i = 0; i < n; ++i) {
        std::vector<int8_t> P, Rp, S;
        encode(prefix[i], P);
        encode(repeat[i], Rp);
        encode(suffix[i], S);
        const int mu = max_units[i];
        LadderDesc& d = lad[i];
        memset(&d, 0, sizeof d);
        if (mu < 0) return fail(c, -2, "ladder %d: negative max_units", i);
        if (mu == 0) {
            if (P.empty() || P.size() > TREDGPU_MAX_TEMPLATE_LEN)
                return fail(c, -2, "ladder %d: reference length %zu not in [1,%d]", i, P.size(), TREDGPU_MAX_TEMPLATE_LEN);
            d.alen[0] = (int)P.size();
            d.trunk_off[0] = append(P);
            d.branch_off[0] = append(std::vector<int8_t>());
            d.period = 1;
            d.max_units = 0;
            d.n_strands = 1;
            continue;
        }
        if (Rp.empty()) return fail(c, -2, "ladder %d: empty repeat", i);
        const size_t T = P.size() + S.size() + Rp.size() * (size_t)mu;
        if (T > TREDGPU_MAX_TEMPLATE_LEN)
            return fail(c, -2, "ladder %d: longest template %zu exceeds %d", i, T, TREDGPU_MAX_TEMPLATE_LEN);
        const std::vector<int8_t> Pr = revcomp(P), Rr = revcomp(Rp), Sr = revcomp(S);
        const std::vector<int8_t>* A[2] = {&P, &Sr};
        const std::vector<int8_t>* Rep[2] = {&Rp, &Rr};
        const std::vector<int8_t>* B[2] = {&S, &Pr};
        for (int s = 0; s < 2; ++s) {
            d.alen[s] = (int)A[s]->size();
            d.blen[s] = (int)B[s]->size();
            std::vector<int8_t> trunk(*A[s]);
            for (int k = 0; k < mu; ++k) trunk.insert(trunk.end(), Rep[s]->begin(), Rep[s]->end());
            d.trunk_off[s] = append(trunk);
            d.branch_off[s] = append(*B[s]);
        }
        d.period = (int)Rp.size();
        d.max_units = mu;
        d.n_strands = 2;
        // 6-mer presence bitmaps over ALL templates of each strand (exact strand filter in the kernel).  A template
        // N scores 0 against every base: it neither breaks nor feeds a run of matches, so a window with N
        // positions stands for all its fillings (the bound's run-length argument then holds unchanged).
        d.kmer_ok = 1;
        for (int s = 0; s < 2; ++s) {
            std::vector<uint32_t> bits(128, 0u);
            for (int u = 1; u <= mu; ++u) {
                std::vector<int8_t> t(*A[s]);
                for (int k = 0; k < u; ++k) t.insert(t.end(), Rep[s]->begin(), Rep[s]->end());
                t.insert(t.end(), B[s]->begin(), B[s]->end());
                for (size_t i = 0; i + 6 <= t.size(); ++i) {
                    uint32_t code = 0;
                    int wild[6], nw = 0;
                    for (int k = 0; k < 6; ++k) {
                        if (t[i + k] > 3) wild[nw++] = k;
                        else code |= (uint32_t)(t[i + k] & 3) << (2 * k);
                    }
                    for (uint32_t f = 0; f < (1u << (2 * nw)); ++f) {
                        uint32_t cf = code;
                        for (int j = 0; j < nw; ++j) cf |= ((f >> (2 * j)) & 3u) << (2 * wild[j]);
                        bits[cf >> 5] |= 1u << (cf & 31);
                    }
                }
            }
            d.kmer_off[s] = (int)seq.size();
            seq.insert(seq.end(), bits.begin(), bits.end());
        }
        max_t = std::max(max_t, 2 * mu);
        max_u = std::max(max_u, mu);
    }
    seq.resize(seq.size() + 4, 0x44444444u);
    int rc;
    if ((rc = ensure(c, c->d_ladders, lad.size() * sizeof(LadderDesc)))) return rc;
    if ((rc = ensure(c, c->d_seq, seq.size() * sizeof(uint32_t)))) return rc;
    HIPCHK(c, stream_sync(c));
    if (n) HIPCHK(c, hipMemcpy(c->d_ladders.p, lad.data(), lad.size() * sizeof(LadderDesc), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_seq.p, seq.data(), seq.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->h_ladders.swap(lad);
    c->max_templates = max_t;
    c->max_ladder_units = max_u;
    return 0;
}

int tredgpu_set_model(tredgpu_ctx* c, const double* step_pdf, const double* stutter_w, double gc, double score) {
    if (!c) return -2;
    if (!step_pdf || !stutter_w) return fail(c, -2, "model arrays are NULL");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<ModelConst> mbuf(1);   // 33 KB: not on the stack
    ModelConst& m = mbuf[0];
    memcpy(m.step, step_pdf, sizeof m.step);
    memcpy(m.w, stutter_w, sizeof m.w);
    m.gc = gc;
    m.score = score;
    m.small = std::exp(-10.0);
    m.really_small = std::exp(-100.0);
    m.logsmall = std::log(m.small);
    for (int n = 0; n < GRID_LFACT; ++n) m.lfact[n] = std::lgamma((double)n + 1);
    int rc;
    if ((rc = ensure(c, c->d_model, sizeof m))) return rc;
    HIPCHK(c, stream_sync(c));
    HIPCHK(c, hipMemcpy(c->d_model.p, &m, sizeof m, hipMemcpyHostToDevice));
    c->have_model = true;
    return 0;
}

int64_t tredgpu_pack_reads(const char* seqs, const int64_t* seq_off, int64_t n_reads, uint32_t* packed_out,
                           int64_t* word_off_out, int32_t* len_out) {
    if (n_reads < 0 || (n_reads > 0 && (!seqs || !seq_off))) return -2;
    int64_t w = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t L = seq_off[r + 1] - seq_off[r];
        if (L < 0 || L > 0x7fffffff) return -2;
        const int64_t nb = (L + 15) >> 4, nm = (L + 31) >> 5;
        if (word_off_out) word_off_out[r] = w;
        if (len_out) len_out[r] = (int32_t)L;
        if (packed_out) {
            uint32_t* rec = packed_out + w;
            for (int64_t k = 0; k < nb + nm; ++k) rec[k] = 0;
            const char* s = seqs + seq_off[r];
            for (int64_t i = 0; i < L; ++i) {
                const int code = base_code(s[i]);
                if (code == 4) rec[nb + (i >> 5)] |= 1u << (i & 31);
                else rec[i >> 4] |= (uint32_t)code << ((i & 15) * 2);
            }
        }
        w += nb + nm;
    }
    if (word_off_out) word_off_out[n_reads] = w;
    return w;
}

// ---- SW ------------------------------------------------------------------------------------------

static int run_sw_device(tredgpu_ctx* c, const uint32_t* packed, const int64_t* read_off, const int32_t* read_len,
                         int64_t n_reads, const int32_t* unit_read_off, const int32_t* unit_ladder,
                         int32_t n_units, const tredgpu_sw_params* p, int max_len, uint8_t* out_tag,
                         int16_t* out_h, int16_t* out_score, int16_t* out_dump, int32_t dump_templates) {
    if (n_reads == 0 || n_units == 0) return 0;
    const int n_ladders = (int)c->h_ladders.size();
    const int64_t max_quads = sw_max_quads(n_reads, n_ladders);
    int rc;
    if ((rc = ensure(c, c->ws_quads, (size_t)max_quads * sizeof(Quad)))) return rc;
    if ((rc = ensure(c, c->ws_counter, 64))) return rc;
    SwArgs a;
    a.packed = packed;
    a.read_off = read_off;
    a.read_len = read_len;
    a.unit_read_off = unit_read_off;
    a.unit_ladder = unit_ladder;
    a.ladders = (const LadderDesc*)c->d_ladders.p;
    a.seqw = (const uint32_t*)c->d_seq.p;
    a.quads = (const Quad*)c->ws_quads.p;
    a.n_quads = (const int32_t*)c->ws_counter.p;
    a.out_tag = out_tag;
    a.out_h = out_h;
    a.out_score = out_score;
    a.out_dump = out_dump;
    a.dump_templates = dump_templates;
    a.n_units = n_units;
    a.n_ladders = n_ladders;
    a.max_rows = 16 * rows_for(max_len);
    a.p = *p;
    a.stats = nullptr;
    if ((rc = ensure(c, c->ws_perm, (size_t)n_reads * sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->ws_class, (size_t)n_reads))) return rc;
    a.perm = (const int32_t*)c->ws_perm.p;
    if ((rc = ensure(c, c->ws_ucnt, sw_unit_cnt_bytes(n_units)))) return rc;
    if ((rc = ensure(c, c->ws_bins, sw_bin_bytes(n_ladders)))) return rc;
    HIPCHK(c, launch_build_quads(a, (uint8_t*)c->ws_class.p, (int32_t*)c->ws_perm.p, (Quad*)c->ws_quads.p,
                                 (int32_t*)c->ws_counter.p, (int32_t*)c->ws_ucnt.p, (int32_t*)c->ws_bins.p, n_ladders,
                                 max_quads, c->stream));
    if (c->ws_stats.p == nullptr) {
        if ((rc = ensure(c, c->ws_stats, SW_STAT_SLOTS * 8 * sizeof(unsigned long long)))) return rc;
        HIPCHK(c, hipMemsetAsync(c->ws_stats.p, 0, SW_STAT_SLOTS * 8 * sizeof(unsigned long long), c->stream));
    }
    a.stats = (unsigned long long*)c->ws_stats.p;
    {
        ScopedTimer tm(c, TREDGPU_KERNEL_SW);
        // a branch (suffix, or prefix on the reverse strand) long enough to pass the score filter of 30 by itself
        // needs the kernel variant that also sweeps the branch alone
        bool generic = false;
        for (const LadderDesc& d : c->h_ladders)
            for (int s2 = 0; s2 < d.n_strands; ++s2) generic = generic || (d.max_units > 0 && d.blen[s2] * p->match >= 30);
        HIPCHK(c, launch_sw_ladder(a, rows_for(max_len), generic, max_quads, c->stream));
    }
    return 0;
}

int tredgpu_sw_classify(tredgpu_ctx* c, int mem, const uint32_t* packed, const int64_t* read_off,
                        const int32_t* read_len, int64_t n_reads, const int32_t* unit_read_off,
                        const int32_t* unit_ladder, int32_t n_units, const tredgpu_sw_params* params,
                        uint8_t* out_tag, int16_t* out_h, int16_t* out_score, int16_t* out_dump,
                        int32_t dump_templates) {
    if (!c) return -2;
    int rc;
    if ((rc = check_sw_params(c, params))) return rc;
    if (n_reads < 0 || n_units < 0) return fail(c, -2, "negative sizes");
    if (n_reads > 0 && (!packed || !read_off || !read_len || !unit_read_off || !unit_ladder || !out_tag || !out_h || !out_score))
        return fail(c, -2, "NULL array argument");
    if (c->h_ladders.empty()) return fail(c, -4, "no ladders registered (tredgpu_set_ladders)");
    if (out_dump && dump_templates <= 0) return fail(c, -2, "dump_templates must be > 0 with out_dump");
    HIPCHK(c, hipSetDevice(c->device));
    int max_len = params->max_read_len;
    if (mem == TREDGPU_MEM_DEVICE) {
        if (max_len <= 0) max_len = TREDGPU_ASSUMED_READ_LEN;
        if (out_dump) HIPCHK(c, hipMemsetAsync(out_dump, 0xFF, (size_t)n_reads * dump_templates * 6 * sizeof(int16_t), c->stream));
        return run_sw_device(c, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, n_units, params,
                             max_len, out_tag, out_h, out_score, out_dump, dump_templates);
    }
    if (mem != TREDGPU_MEM_HOST) return fail(c, -2, "mem must be TREDGPU_MEM_HOST or TREDGPU_MEM_DEVICE");
    if (n_reads == 0) return 0;
    // validate the host metadata (the device path trusts its caller)
    if (unit_read_off[0] != 0 || unit_read_off[n_units] != n_reads) return fail(c, -2, "unit_read_off must span [0,n_reads]");
    for (int g = 0; g < n_units; ++g) {
        if (unit_read_off[g + 1] < unit_read_off[g]) return fail(c, -2, "unit_read_off not monotone at %d", g);
        if (unit_ladder[g] < 0 || unit_ladder[g] >= (int)c->h_ladders.size()) return fail(c, -2, "unit %d: ladder %d not registered", g, unit_ladder[g]);
    }
    int seen = 0;
    for (int64_t r = 0; r < n_reads; ++r) seen = std::max(seen, read_len[r]);
    if (max_len <= 0) max_len = seen;
    if (seen > TREDGPU_MAX_READ_LEN) return fail(c, -5, "read of %d bp exceeds TREDGPU_MAX_READ_LEN=%d", seen, TREDGPU_MAX_READ_LEN);
    if ((rc = check_sw_range(c, params, std::max(max_len, 1)))) return rc;     // the instantiation these reads select
    const uint32_t* d_packed; const int64_t* d_off; const int32_t* d_len; const int32_t* d_uoff; const int32_t* d_ulad;
    uint8_t* d_tag; int16_t* d_h; int16_t* d_score; int16_t* d_dump = nullptr;
    const size_t words = (size_t)read_off[n_reads];
    if ((rc = stage_in(c, c->st[0], packed, words, &d_packed))) return rc;
    if ((rc = stage_in(c, c->st[1], read_off, (size_t)n_reads + 1, &d_off))) return rc;
    if ((rc = stage_in(c, c->st[2], read_len, (size_t)n_reads, &d_len))) return rc;
    if ((rc = stage_in(c, c->st[3], unit_read_off, (size_t)n_units + 1, &d_uoff))) return rc;
    if ((rc = stage_in(c, c->st[4], unit_ladder, (size_t)n_units, &d_ulad))) return rc;
    if ((rc = stage_out(c, c->st[5], (size_t)n_reads, &d_tag))) return rc;
    if ((rc = stage_out(c, c->st[6], (size_t)n_reads, &d_h))) return rc;
    if ((rc = stage_out(c, c->st[7], (size_t)n_reads, &d_score))) return rc;
    const size_t dump_n = out_dump ? (size_t)n_reads * dump_templates * 6 : 0;
    if (out_dump) {
        if ((rc = stage_out(c, c->st[8], dump_n, &d_dump))) return rc;
        HIPCHK(c, hipMemsetAsync(d_dump, 0xFF, dump_n * sizeof(int16_t), c->stream));
    }
    if ((rc = run_sw_device(c, d_packed, d_off, d_len, n_reads, d_uoff, d_ulad, n_units, params, max_len, d_tag,
                            d_h, d_score, d_dump, dump_templates)))
        return rc;
    if ((rc = copy_back(c, out_tag, (const uint8_t*)d_tag, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_h, (const int16_t*)d_h, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_score, (const int16_t*)d_score, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_dump, (const int16_t*)d_dump, dump_n))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

int tredgpu_tally(tredgpu_ctx* c, int mem, const uint8_t* tag, const int16_t* h, int64_t n_reads,
                  const int32_t* unit_read_off, int32_t n_units, const int32_t* read_pair_id, int32_t hist_stride,
                  int32_t* full_cnt, int32_t* pref_cnt, int32_t* rept_cnt) {
    if (!c) return -2;
    if (n_reads < 0 || n_units < 0 || hist_stride <= 0) return fail(c, -2, "bad sizes");
    if (n_units > 0 && (!unit_read_off || !full_cnt || !pref_cnt || !rept_cnt)) return fail(c, -2, "NULL array argument");
    if (n_reads > 0 && (!tag || !h)) return fail(c, -2, "NULL array argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (read_pair_id && (rc = ensure(c, c->ws_drop, (size_t)n_reads))) return rc;
    if (mem == TREDGPU_MEM_DEVICE) {
        ScopedTimer tm(c, TREDGPU_KERNEL_TALLY);
        HIPCHK(c, launch_tally(tag, h, n_reads, unit_read_off, n_units, read_pair_id, hist_stride, full_cnt, pref_cnt,
                               rept_cnt, (uint8_t*)c->ws_drop.p, c->stream));
        return 0;
    }
    if (mem != TREDGPU_MEM_HOST) return fail(c, -2, "bad mem");
    const uint8_t* d_tag; const int16_t* d_h; const int32_t* d_uoff; const int32_t* d_pid = nullptr;
    int32_t *d_f, *d_p, *d_r;
    const size_t hn = (size_t)n_units * hist_stride;
    if ((rc = stage_in(c, c->st[0], tag, (size_t)n_reads, &d_tag))) return rc;
    if ((rc = stage_in(c, c->st[1], h, (size_t)n_reads, &d_h))) return rc;
    if ((rc = stage_in(c, c->st[2], unit_read_off, (size_t)n_units + 1, &d_uoff))) return rc;
    if (read_pair_id && (rc = stage_in(c, c->st[3], read_pair_id, (size_t)n_reads, &d_pid))) return rc;
    if ((rc = stage_out(c, c->st[4], hn, &d_f))) return rc;
    if ((rc = stage_out(c, c->st[5], hn, &d_p))) return rc;
    if ((rc = stage_out(c, c->st[6], hn, &d_r))) return rc;
    {
        ScopedTimer tm(c, TREDGPU_KERNEL_TALLY);
        HIPCHK(c, launch_tally(d_tag, d_h, n_reads, d_uoff, n_units, d_pid, hist_stride, d_f, d_p, d_r,
                               (uint8_t*)c->ws_drop.p, c->stream));
    }
    if ((rc = copy_back(c, full_cnt, (const int32_t*)d_f, hn))) return rc;
    if ((rc = copy_back(c, pref_cnt, (const int32_t*)d_p, hn))) return rc;
    if ((rc = copy_back(c, rept_cnt, (const int32_t*)d_r, hn))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

// ---- likelihood grid -------------------------------------------------------------------------------

static int check_grid_common(tredgpu_ctx* c, const tredgpu_unit_params* units, int32_t n_units) {
    if (!c->have_model) return fail(c, -4, "model constants not set (tredgpu_set_model)");
    if (n_units < 0) return fail(c, -2, "negative n_units");
    if (n_units > 0 && !units) return fail(c, -2, "units is NULL");
    return 0;
}

// Largest maxinsert and n_target over the units (device array): they bound the grid axes and the paired-end
// tables, hence the scratch a unit can ask for.
// One 4-byte read-back; the fused path asks before it launches the SW kernel, so the stream is idle.
static int query_max_insert(tredgpu_ctx* c, const tredgpu_unit_params* units, int32_t n_units, int out[2]) {
    int rc;
    if ((rc = ensure(c, c->ws_counter, 64))) return rc;
    if (!c->h_pin) HIPCHK(c, hipHostMalloc((void**)&c->h_pin, 64, hipHostMallocDefault));
    int* d = (int32_t*)c->ws_counter.p + 12;
    HIPCHK(c, launch_unit_max(units, n_units, d, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_pin, d, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_sync(c));
    out[0] = c->h_pin[0];   // largest maxinsert
    out[1] = c->h_pin[1];   // largest n_target
    return 0;
}

static int run_grid_device(tredgpu_ctx* c, const tredgpu_unit_params* units, int32_t n_units, int32_t hist_stride,
                           const int32_t* full_cnt, const int32_t* pref_cnt, const int32_t* rept_cnt,
                           const int32_t* global_lens, const int32_t* target_lens, tredgpu_call* calls,
                           const int64_t* grid_off, double* grid_dump, double* marg, int32_t marg_stride,
                           const int* limits /* {max maxinsert, max n_target}; NULL: ask the device */,
                           const int64_t* joint_off = nullptr, double* joint = nullptr, int32_t* joint_n = nullptr,
                           double* joint_total = nullptr) {
    if (n_units == 0) return 0;
    int rc;
    int lim[2];
    if (limits) { lim[0] = limits[0]; lim[1] = limits[1]; }
    else if ((rc = query_max_insert(c, units, n_units, lim))) return rc;
    const int max_insert = lim[0], max_target = lim[1];
    // every axis is a subset of {distinct sizes} U {max_partial} plus at most maxinsert arithmetic entries
    const int cap = std::min(std::max(hist_stride + 1 + std::max(max_insert, 0), 8), GRID_MAX_ROWS);
    const size_t slot_max = grid_slot_doubles_max(cap, cap, max_target) * sizeof(double);
    // the pool: everything the batch can ask for if that is small, else GRID_POOL_BYTES and as many passes
    // as it takes (units that find their sub-pool full are deferred to the next pass).  The kernels use it as up to 16
    // sub-pools, unit g in sub-pool g % n: "everything" = the worst case of the fullest residue class, n times
    const size_t slot_room = slot_max + 16 * sizeof(double);
    const size_t all_bytes = 16 * grid_units_per_subpool(n_units, 16) * slot_room;
    const size_t pool_bytes = std::max(std::min(all_bytes, c->grid_pool_bytes), slot_room);
    const bool may_defer = all_bytes > pool_bytes;
    const size_t item_cap = grid_item_slots(n_units, cap, cap);
    if ((rc = ensure(c, c->ws_grid, pool_bytes))) return rc;
    if ((rc = ensure(c, c->ws_gdesc, (size_t)n_units * grid_desc_bytes()))) return rc;
    if ((rc = ensure(c, c->ws_gtile, item_cap * grid_item_bytes() + 64))) return rc;
    if ((rc = ensure(c, c->ws_gctr, grid_counter_bytes()))) return rc;
    // the units' paired-end KDEs (grid_kde_kernel -> grid_prepare_kernel): 1000 doubles + an outcome per unit
    const size_t kde_pdf_bytes = (size_t)n_units * TREDGPU_SPAN * sizeof(double);
    if ((rc = ensure(c, c->ws_kde, kde_pdf_bytes + (size_t)n_units * sizeof(int32_t)))) return rc;
    if (!c->h_pin) HIPCHK(c, hipHostMalloc((void**)&c->h_pin, 64, hipHostMallocDefault));
    GridArgs a;
    a.units = units;
    a.n_units = n_units;
    a.hist_stride = hist_stride;
    a.full_cnt = full_cnt;
    a.pref_cnt = pref_cnt;
    a.rept_cnt = rept_cnt;
    a.global_lens = global_lens;
    a.target_lens = target_lens;
    a.calls = calls;
    a.grid_off = grid_off;
    a.grid_dump = grid_dump;
    a.marg = marg;
    a.marg_stride = marg_stride;
    a.model = (const ModelConst*)c->d_model.p;
    a.joint_off = joint_off;
    a.joint = joint;
    a.joint_n = joint_n;
    a.joint_total = joint_total;
    a.kde_pdf = nullptr;
    a.kde_status = nullptr;
    a.unit_pdf = (double*)c->ws_kde.p;
    a.unit_kde_rc = (int32_t*)((char*)c->ws_kde.p + kde_pdf_bytes);
    a.max_target = max_target;
    {
        ScopedTimer tm(c, TREDGPU_KERNEL_GRID);
        for (int pass = 0;; ++pass) {
            // KDE (first pass only) / prepare / pairs / reduce, each bracketed by its own events
            static const int PHASE[4][2] = {{8, TREDGPU_KERNEL_GRID_KDE}, {1, TREDGPU_KERNEL_GRID_PREPARE},
                                            {2, TREDGPU_KERNEL_GRID_PAIRS}, {4, TREDGPU_KERNEL_GRID_REDUCE}};
            for (int ph = pass == 0 ? 0 : 1; ph < 4; ++ph) {
                ScopedTimer phase_tm(c, PHASE[ph][1]);
                HIPCHK(c, launch_grid_pass(a, pass, c->ws_gdesc.p, (double*)c->ws_grid.p, pool_bytes / sizeof(double), cap,
                                           cap, c->ws_gtile.p, item_cap, c->ws_gctr.p, c->stream, PHASE[ph][0]));
            }
            if (!may_defer) break;
            HIPCHK(c, hipMemcpyAsync(c->h_pin, (const char*)c->ws_gctr.p + grid_deferred_offset(), sizeof(int),
                                     hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, stream_sync(c));
            if (c->h_pin[0] == 0) break;
            if (pass >= 1000) return fail(c, -10, "likelihood grid: scratch pool passes do not converge");
        }
    }
    return 0;
}

static int likelihood_grid_impl(tredgpu_ctx* c, int mem, const tredgpu_unit_params* units, int32_t n_units,
                            int32_t hist_stride, const int32_t* full_cnt, const int32_t* pref_cnt,
                            const int32_t* rept_cnt, const int32_t* global_lens, int64_t n_global_total,
                            const int32_t* target_lens, int64_t n_target_total, tredgpu_call* calls,
                            const int64_t* grid_off, double* grid_dump, double* marg, int32_t marg_stride,
                            const int64_t* joint_off, double* joint, int32_t* joint_n, double* joint_total) {
    if (!c) return -2;
    int rc;
    if ((rc = check_grid_common(c, units, n_units))) return rc;
    if (hist_stride <= 0) return fail(c, -2, "hist_stride must be > 0");
    if (n_units > 0 && (!full_cnt || !pref_cnt || !rept_cnt || !calls)) return fail(c, -2, "NULL array argument");
    if ((grid_off == nullptr) != (grid_dump == nullptr)) return fail(c, -2, "grid_off and grid_dump go together");
    if (joint_off && (!joint || !joint_n || !joint_total)) return fail(c, -2, "joint_off, joint, joint_n and joint_total go together");
    if (marg && marg_stride <= 0) return fail(c, -2, "marg_stride must be > 0");
    HIPCHK(c, hipSetDevice(c->device));
    if (mem == TREDGPU_MEM_DEVICE)
        return run_grid_device(c, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens, target_lens,
                               calls, grid_off, grid_dump, marg, marg_stride, nullptr, joint_off, joint, joint_n,
                               joint_total);
    if (mem != TREDGPU_MEM_HOST) return fail(c, -2, "bad mem");
    if (n_units == 0) return 0;
    int limits[2] = {0, 0};
    for (int g = 0; g < n_units; ++g) {
        limits[0] = std::max(limits[0], units[g].maxinsert);
        limits[1] = std::max(limits[1], units[g].n_target);
    }
    for (int g = 0; g < n_units; ++g) {
        const tredgpu_unit_params& u = units[g];
        if (u.n_global < 0 || u.n_target < 0 || u.pe_off < 0 || u.tl_off < 0 || (int64_t)u.pe_off + u.n_global > n_global_total ||
            (int64_t)u.tl_off + u.n_target > n_target_total)
            return fail(c, -2, "unit %d: paired-end slices out of range", g);
        if (marg && marg_stride <= std::max(u.maxinsert, hist_stride)) return fail(c, -2, "marg_stride must exceed max(maxinsert, hist_stride)");
    }
    const tredgpu_unit_params* d_units; const int32_t *d_f, *d_p, *d_r, *d_gl = nullptr, *d_tl = nullptr;
    const int64_t* d_goff = nullptr; tredgpu_call* d_calls; double* d_dump = nullptr; double* d_marg = nullptr;
    const size_t hn = (size_t)n_units * hist_stride;
    if ((rc = stage_in(c, c->st[0], units, (size_t)n_units, &d_units))) return rc;
    if ((rc = stage_in(c, c->st[1], full_cnt, hn, &d_f))) return rc;
    if ((rc = stage_in(c, c->st[2], pref_cnt, hn, &d_p))) return rc;
    if ((rc = stage_in(c, c->st[3], rept_cnt, hn, &d_r))) return rc;
    if ((rc = stage_in(c, c->st[4], global_lens, (size_t)n_global_total, &d_gl))) return rc;
    if ((rc = stage_in(c, c->st[5], target_lens, (size_t)n_target_total, &d_tl))) return rc;
    if ((rc = stage_out(c, c->st[6], (size_t)n_units, &d_calls))) return rc;
    size_t dump_n = 0;
    if (grid_off) {
        dump_n = (size_t)grid_off[n_units] * 6;
        if ((rc = stage_in(c, c->st[7], grid_off, (size_t)n_units + 1, &d_goff))) return rc;
        if ((rc = stage_out(c, c->st[8], dump_n, &d_dump))) return rc;
    }
    const size_t marg_n = marg ? (size_t)n_units * 2 * marg_stride : 0;
    if (marg && (rc = stage_out(c, c->st[9], marg_n, &d_marg))) return rc;
    const int64_t* d_joff = nullptr; double* d_joint = nullptr; int32_t* d_jn = nullptr; double* d_jt = nullptr;
    size_t joint_len = 0;
    if (joint_off) {
        joint_len = (size_t)joint_off[n_units] * 3;
        if ((rc = stage_in(c, c->st[10], joint_off, (size_t)n_units + 1, &d_joff))) return rc;
        if ((rc = stage_out(c, c->st[11], joint_len, &d_joint))) return rc;
        if ((rc = stage_out(c, c->st[12], (size_t)n_units, &d_jn))) return rc;
        if ((rc = stage_out(c, c->st[13], (size_t)n_units, &d_jt))) return rc;
    }
    if ((rc = run_grid_device(c, d_units, n_units, hist_stride, d_f, d_p, d_r, d_gl, d_tl, d_calls, d_goff, d_dump,
                              d_marg, marg_stride, limits, d_joff, d_joint, d_jn, d_jt)))
        return rc;
    if ((rc = copy_back(c, joint, (const double*)d_joint, joint_len))) return rc;
    if ((rc = copy_back(c, joint_n, (const int32_t*)d_jn, joint_off ? (size_t)n_units : 0))) return rc;
    if ((rc = copy_back(c, joint_total, (const double*)d_jt, joint_off ? (size_t)n_units : 0))) return rc;
    if ((rc = copy_back(c, calls, (const tredgpu_call*)d_calls, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, grid_dump, (const double*)d_dump, dump_n))) return rc;
    if ((rc = copy_back(c, marg, (const double*)d_marg, marg_n))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

int tredgpu_likelihood_grid(tredgpu_ctx* c, int mem, const tredgpu_unit_params* units, int32_t n_units,
                            int32_t hist_stride, const int32_t* full_cnt, const int32_t* pref_cnt,
                            const int32_t* rept_cnt, const int32_t* global_lens, int64_t n_global_total,
                            const int32_t* target_lens, int64_t n_target_total, tredgpu_call* calls,
                            const int64_t* grid_off, double* grid_dump, double* marg, int32_t marg_stride) {
    return likelihood_grid_impl(c, mem, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens,
                                n_global_total, target_lens, n_target_total, calls, grid_off, grid_dump, marg,
                                marg_stride, nullptr, nullptr, nullptr, nullptr);
}

int tredgpu_likelihood_grid_joint(tredgpu_ctx* c, int mem, const tredgpu_unit_params* units, int32_t n_units,
                                  int32_t hist_stride, const int32_t* full_cnt, const int32_t* pref_cnt,
                                  const int32_t* rept_cnt, const int32_t* global_lens, int64_t n_global_total,
                                  const int32_t* target_lens, int64_t n_target_total, tredgpu_call* calls,
                                  double* marg, int32_t marg_stride, const int64_t* joint_off, double* joint,
                                  int32_t* joint_n, double* joint_total) {
    if (c && !joint_off) return fail(c, -2, "joint_off is NULL");
    return likelihood_grid_impl(c, mem, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens,
                                n_global_total, target_lens, n_target_total, calls, nullptr, nullptr, marg,
                                marg_stride, joint_off, joint, joint_n, joint_total);
}

int tredgpu_pe_kde(tredgpu_ctx* c, int mem, const tredgpu_unit_params* units, int32_t n_units,
                   const int32_t* global_lens, int64_t n_global_total, double* pdf_out, int32_t* status_out) {
    if (!c) return -2;
    if (n_units < 0 || (n_units > 0 && (!units || !pdf_out || !status_out))) return fail(c, -2, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (n_units == 0) return 0;
    GridArgs a;
    memset(&a, 0, sizeof a);
    a.n_units = n_units;
    int rc;
    if (mem == TREDGPU_MEM_DEVICE) {
        a.units = units;
        a.global_lens = global_lens;
        a.kde_pdf = pdf_out;
        a.kde_status = status_out;
        HIPCHK(c, launch_pe_kde(a, c->stream));
        return 0;
    }
    if (mem != TREDGPU_MEM_HOST) return fail(c, -2, "bad mem");
    const tredgpu_unit_params* d_units; const int32_t* d_gl; double* d_pdf; int32_t* d_st;
    if ((rc = stage_in(c, c->st[0], units, (size_t)n_units, &d_units))) return rc;
    if ((rc = stage_in(c, c->st[1], global_lens, (size_t)n_global_total, &d_gl))) return rc;
    if ((rc = stage_out(c, c->st[2], (size_t)n_units * TREDGPU_SPAN, &d_pdf))) return rc;
    if ((rc = stage_out(c, c->st[3], (size_t)n_units, &d_st))) return rc;
    a.units = d_units;
    a.global_lens = d_gl;
    a.kde_pdf = d_pdf;
    a.kde_status = d_st;
    HIPCHK(c, launch_pe_kde(a, c->stream));
    if ((rc = copy_back(c, pdf_out, (const double*)d_pdf, (size_t)n_units * TREDGPU_SPAN))) return rc;
    if ((rc = copy_back(c, status_out, (const int32_t*)d_st, (size_t)n_units))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

// The whole path for a batch in HOST memory with everything the product's writers print, in ONE call with ONE wait:
// tredgpu_sw_classify -> tredgpu_tally -> tredgpu_likelihood_grid_joint composed on the device -- tags, histograms and
// calls never leave it between the stages (the three separate calls cost a driver three waits and two copies of the
// per-read arrays in each direction: 20 ms per batch of 16 samples, most of it waiting among other processes' work).
int tredgpu_genotype_batch_joint(tredgpu_ctx* c, const uint32_t* packed, const int64_t* read_off, const int32_t* read_len,
                                 int64_t n_reads, const int32_t* unit_read_off, const int32_t* unit_ladder,
                                 const tredgpu_unit_params* units, int32_t n_units, const tredgpu_sw_params* params,
                                 const int32_t* read_pair_id, const int32_t* global_lens, int64_t n_global_total,
                                 const int32_t* target_lens, int64_t n_target_total, uint8_t* out_tag, int16_t* out_h,
                                 int16_t* out_score, int32_t hist_stride, int32_t* rept_cnt, tredgpu_call* calls, double* marg,
                                 int32_t marg_stride, const int64_t* joint_off, double* joint, int32_t* joint_n,
                                 double* joint_total) {
    if (!c) return -2;
    int rc;
    if ((rc = check_sw_params(c, params))) return rc;
    if ((rc = check_grid_common(c, units, n_units))) return rc;
    if (c->h_ladders.empty()) return fail(c, -4, "no ladders registered (tredgpu_set_ladders)");
    if (n_reads < 0 || n_units <= 0) return fail(c, -2, "bad sizes");
    if (!unit_read_off || !unit_ladder || !calls || !marg || !joint_off || !joint || !joint_n || !joint_total || marg_stride <= 0)
        return fail(c, -2, "NULL array argument");
    if (n_reads > 0 && (!packed || !read_off || !read_len || !out_tag || !out_h || !out_score)) return fail(c, -2, "NULL array argument");
    HIPCHK(c, hipSetDevice(c->device));
    // validate the host metadata (as the separate host-memory calls do)
    if (unit_read_off[0] != 0 || unit_read_off[n_units] != n_reads) return fail(c, -2, "unit_read_off must span [0,n_reads]");
    int limits[2] = {0, 0};
    for (int g = 0; g < n_units; ++g) {
        if (unit_read_off[g + 1] < unit_read_off[g]) return fail(c, -2, "unit_read_off not monotone at %d", g);
        if (unit_ladder[g] < 0 || unit_ladder[g] >= (int)c->h_ladders.size()) return fail(c, -2, "unit %d: ladder %d not registered", g, unit_ladder[g]);
        // (the histograms must hold the repeat counts of THIS batch's ladders; the context may know longer ones)
        if (hist_stride <= c->h_ladders[unit_ladder[g]].max_units)
            return fail(c, -2, "hist_stride %d must exceed the max_units %d of unit %d's ladder", hist_stride, c->h_ladders[unit_ladder[g]].max_units, g);
        const tredgpu_unit_params& u = units[g];
        limits[0] = std::max(limits[0], u.maxinsert);
        limits[1] = std::max(limits[1], u.n_target);
        if (u.n_global < 0 || u.n_target < 0 || u.pe_off < 0 || u.tl_off < 0 || (int64_t)u.pe_off + u.n_global > n_global_total ||
            (int64_t)u.tl_off + u.n_target > n_target_total)
            return fail(c, -2, "unit %d: paired-end slices out of range", g);
        if (marg_stride <= std::max(u.maxinsert, hist_stride)) return fail(c, -2, "marg_stride must exceed max(maxinsert, hist_stride)");
    }
    int seen = 0;
    for (int64_t r = 0; r < n_reads; ++r) seen = std::max(seen, read_len[r]);
    if (seen > TREDGPU_MAX_READ_LEN) return fail(c, -5, "read of %d bp exceeds TREDGPU_MAX_READ_LEN=%d", seen, TREDGPU_MAX_READ_LEN);
    const int max_len = params->max_read_len > 0 ? params->max_read_len : std::max(seen, 1);
    if ((rc = check_sw_range(c, params, max_len))) return rc;                  // the instantiation these reads select
    const uint32_t* d_packed = nullptr; const int64_t* d_off = nullptr; const int32_t* d_len = nullptr;
    const int32_t *d_uoff, *d_ulad, *d_pid = nullptr, *d_gl = nullptr, *d_tl = nullptr;
    const tredgpu_unit_params* d_units; const int64_t* d_joff;
    uint8_t* d_tag; int16_t *d_h, *d_score; int32_t *d_f, *d_p, *d_r; tredgpu_call* d_calls; double *d_marg, *d_joint, *d_jt; int32_t* d_jn;
    const size_t words = n_reads > 0 ? (size_t)read_off[n_reads] : 0, hn = (size_t)n_units * hist_stride;
    const size_t marg_n = (size_t)n_units * 2 * marg_stride, joint_len = (size_t)joint_off[n_units] * 3;
    if ((rc = stage_in(c, c->st[0], packed, words, &d_packed))) return rc;
    if ((rc = stage_in(c, c->st[1], read_off, n_reads > 0 ? (size_t)n_reads + 1 : 0, &d_off))) return rc;
    if ((rc = stage_in(c, c->st[2], read_len, (size_t)n_reads, &d_len))) return rc;
    if ((rc = stage_in(c, c->st[3], unit_read_off, (size_t)n_units + 1, &d_uoff))) return rc;
    if ((rc = stage_in(c, c->st[4], unit_ladder, (size_t)n_units, &d_ulad))) return rc;
    if ((rc = stage_out(c, c->st[5], (size_t)n_reads, &d_tag))) return rc;
    if ((rc = stage_out(c, c->st[6], (size_t)n_reads, &d_h))) return rc;
    if ((rc = stage_out(c, c->st[7], (size_t)n_reads, &d_score))) return rc;
    if (read_pair_id && (rc = stage_in(c, c->st[8], read_pair_id, (size_t)n_reads, &d_pid))) return rc;
    if (read_pair_id && (rc = ensure(c, c->ws_drop, (size_t)n_reads))) return rc;
    if ((rc = stage_out(c, c->st[14], hn, &d_f))) return rc;
    if ((rc = stage_out(c, c->st[15], hn, &d_p))) return rc;
    if ((rc = stage_out(c, c->st[16], hn, &d_r))) return rc;
    if ((rc = stage_in(c, c->st[17], units, (size_t)n_units, &d_units))) return rc;
    if ((rc = stage_in(c, c->st[18], global_lens, (size_t)n_global_total, &d_gl))) return rc;
    if ((rc = stage_in(c, c->st[19], target_lens, (size_t)n_target_total, &d_tl))) return rc;
    if ((rc = stage_out(c, c->st[20], (size_t)n_units, &d_calls))) return rc;
    if ((rc = stage_out(c, c->st[21], marg_n, &d_marg))) return rc;
    if ((rc = stage_in(c, c->st[22], joint_off, (size_t)n_units + 1, &d_joff))) return rc;
    if ((rc = stage_out(c, c->st[23], joint_len, &d_joint))) return rc;
    if ((rc = stage_out(c, c->st[24], (size_t)n_units, &d_jn))) return rc;
    if ((rc = stage_out(c, c->st[25], (size_t)n_units, &d_jt))) return rc;
    if (n_reads > 0 && (rc = run_sw_device(c, d_packed, d_off, d_len, n_reads, d_uoff, d_ulad, n_units, params, max_len, d_tag, d_h,
                                           d_score, nullptr, 0)))
        return rc;
    {
        ScopedTimer tm(c, TREDGPU_KERNEL_TALLY);
        HIPCHK(c, launch_tally(d_tag, d_h, n_reads, d_uoff, n_units, d_pid, hist_stride, d_f, d_p, d_r, (uint8_t*)c->ws_drop.p, c->stream));
    }
    if ((rc = run_grid_device(c, d_units, n_units, hist_stride, d_f, d_p, d_r, d_gl, d_tl, d_calls, nullptr, nullptr, d_marg,
                              marg_stride, limits, d_joff, d_joint, d_jn, d_jt)))
        return rc;
    if ((rc = copy_back(c, out_tag, (const uint8_t*)d_tag, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_h, (const int16_t*)d_h, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_score, (const int16_t*)d_score, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, rept_cnt, (const int32_t*)d_r, rept_cnt ? hn : 0))) return rc;
    if ((rc = copy_back(c, joint, (const double*)d_joint, joint_len))) return rc;
    if ((rc = copy_back(c, joint_n, (const int32_t*)d_jn, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, joint_total, (const double*)d_jt, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, calls, (const tredgpu_call*)d_calls, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, marg, (const double*)d_marg, marg_n))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

// tredgpu_genotype_batch_joint over reads that never left the device (include/tredgpu.h section 5): the units are tasks of
// inflaters' selections; pack_selected_kernel writes them into this context's buffers in the layout the SW kernel reads, and
// the selected reads' lengths, 4-bit sequences and names come back with the results -- one wait for all of it.
int tredgpu_genotype_selected(tredgpu_ctx* c, const tredgpu_selected_units* segs, int32_t n_segs, const int32_t* unit_read_off,
                              const int64_t* unit_word_off, const int64_t* unit_seq4_off, const int64_t* unit_name_off,
                              const int32_t* unit_ladder, const tredgpu_unit_params* units, int32_t n_units,
                              const tredgpu_sw_params* params, const int32_t* global_lens, int64_t n_global_total,
                              const int32_t* target_lens, int64_t n_target_total, uint8_t* out_tag, int16_t* out_h,
                              int16_t* out_score, int32_t hist_stride, int32_t* rept_cnt, tredgpu_call* calls, double* marg,
                              int32_t marg_stride, const int64_t* joint_off, double* joint, int32_t* joint_n, double* joint_total,
                              int32_t* read_len, int64_t* seq4_off, uint8_t* seq4, int64_t* name_off, char* names) {
    if (!c) return -2;
    int rc;
    if ((rc = check_sw_params(c, params))) return rc;
    if ((rc = check_grid_common(c, units, n_units))) return rc;
    if (c->h_ladders.empty()) return fail(c, -4, "no ladders registered (tredgpu_set_ladders)");
    if (n_units <= 0 || n_segs <= 0 || !segs) return fail(c, -2, "bad sizes");
    if (!unit_read_off || !unit_word_off || !unit_seq4_off || !unit_name_off || !unit_ladder || !calls || !marg || !joint_off || !joint || !joint_n ||
        !joint_total || marg_stride <= 0)
        return fail(c, -2, "NULL array argument");
    if (params->max_read_len <= 0 || params->max_read_len > TREDGPU_MAX_READ_LEN) return fail(c, -2, "params.max_read_len must name the longest selected read");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n_reads = unit_read_off[n_units];
    if (unit_read_off[0] != 0 || unit_word_off[0] != 0 || unit_seq4_off[0] != 0 || unit_name_off[0] != 0 || n_reads < 0) return fail(c, -2, "unit offsets must start at 0");
    if (n_reads > 0 && (!out_tag || !out_h || !out_score || !read_len || !seq4_off || !seq4 || !name_off || !names)) return fail(c, -2, "NULL array argument");
    // the units against the inflaters' own records of what they selected
    std::vector<int32_t> unit_task((size_t)n_units);
    std::vector<tredgpu_front::SelectedView> views((size_t)n_segs);
    int limits[2] = {0, 0};
    int g = 0;
    for (int32_t sgi = 0; sgi < n_segs; ++sgi) {
        const tredgpu_selected_units& sg = segs[sgi];
        if (!sg.inf || sg.n_units < 0 || (sg.n_units > 0 && !sg.task) || g + sg.n_units > n_units) return fail(c, -2, "segment %d: bad unit list", (int)sgi);
        if (tredgpu_front::inflater_selected(sg.inf, &views[sgi]) != 0) return fail(c, -2, "segment %d: the inflater's last call carried no read selection", (int)sgi);
        if (views[sgi].device != c->device) return fail(c, -2, "segment %d: the inflater lives on another device", (int)sgi);
        for (int32_t k = 0; k < sg.n_units; ++k, ++g) {
            const int32_t t = sg.task[k];
            if (t < 0 || t >= views[sgi].n_tasks) return fail(c, -2, "unit %d: task %d is none of the inflater's", g, (int)t);
            const tredgpu_select_result& R = views[sgi].results[t];
            if (R.status != 0) return fail(c, -2, "unit %d: its selection was declined (status %d)", g, (int)R.status);
            if (unit_read_off[g + 1] - unit_read_off[g] != R.n_reads || unit_word_off[g + 1] - unit_word_off[g] != R.n_words ||
                unit_seq4_off[g + 1] - unit_seq4_off[g] != R.seq4_bytes || unit_name_off[g + 1] - unit_name_off[g] != R.name_bytes)
                return fail(c, -2, "unit %d: offsets do not match what was selected", g);
            if (R.max_len > params->max_read_len) return fail(c, -5, "unit %d: a read of %d bp exceeds params.max_read_len", g, (int)R.max_len);
            unit_task[(size_t)g] = t;
        }
    }
    if (g != n_units) return fail(c, -2, "the segments hold %d units, the batch %d", g, (int)n_units);
    for (int u = 0; u < n_units; ++u) {
        if (unit_ladder[u] < 0 || unit_ladder[u] >= (int)c->h_ladders.size()) return fail(c, -2, "unit %d: ladder %d not registered", u, unit_ladder[u]);
        if (hist_stride <= c->h_ladders[unit_ladder[u]].max_units)
            return fail(c, -2, "hist_stride %d must exceed the max_units %d of unit %d's ladder", hist_stride, c->h_ladders[unit_ladder[u]].max_units, u);
        const tredgpu_unit_params& up = units[u];
        limits[0] = std::max(limits[0], up.maxinsert);
        limits[1] = std::max(limits[1], up.n_target);
        if (up.n_global < 0 || up.n_target < 0 || up.pe_off < 0 || up.tl_off < 0 || (int64_t)up.pe_off + up.n_global > n_global_total ||
            (int64_t)up.tl_off + up.n_target > n_target_total)
            return fail(c, -2, "unit %d: paired-end slices out of range", u);
        if (marg_stride <= std::max(up.maxinsert, hist_stride)) return fail(c, -2, "marg_stride must exceed max(maxinsert, hist_stride)");
    }
    const int max_len = params->max_read_len;
    const size_t words = (size_t)unit_word_off[n_units], s4_bytes = (size_t)unit_seq4_off[n_units], nm_bytes = (size_t)unit_name_off[n_units];
    const size_t hn = (size_t)n_units * hist_stride, marg_n = (size_t)n_units * 2 * marg_stride, joint_len = (size_t)joint_off[n_units] * 3;
    uint32_t* d_packed; int64_t* d_off; int32_t* d_len; uint8_t* d_seq4; int64_t* d_s4off; uint8_t* d_names; int64_t* d_nmoff;
    const int32_t *d_uoff, *d_ulad, *d_task, *d_gl = nullptr, *d_tl = nullptr;
    const int64_t *d_uw, *d_us, *d_un, *d_joff;
    const tredgpu_unit_params* d_units;
    uint8_t* d_tag; int16_t *d_h, *d_score; int32_t *d_f, *d_p, *d_r; tredgpu_call* d_calls; double *d_marg, *d_joint, *d_jt; int32_t* d_jn;
    if ((rc = stage_out(c, c->st[0], words, &d_packed))) return rc;
    if ((rc = stage_out(c, c->st[1], (size_t)n_reads + 1, &d_off))) return rc;
    if ((rc = stage_out(c, c->st[2], (size_t)n_reads, &d_len))) return rc;
    if ((rc = stage_in(c, c->st[3], unit_read_off, (size_t)n_units + 1, &d_uoff))) return rc;
    if ((rc = stage_in(c, c->st[4], unit_ladder, (size_t)n_units, &d_ulad))) return rc;
    if ((rc = stage_out(c, c->st[5], (size_t)n_reads, &d_tag))) return rc;
    if ((rc = stage_out(c, c->st[6], (size_t)n_reads, &d_h))) return rc;
    if ((rc = stage_out(c, c->st[7], (size_t)n_reads, &d_score))) return rc;
    if ((rc = stage_out(c, c->st[14], hn, &d_f))) return rc;
    if ((rc = stage_out(c, c->st[15], hn, &d_p))) return rc;
    if ((rc = stage_out(c, c->st[16], hn, &d_r))) return rc;
    if ((rc = stage_in(c, c->st[17], units, (size_t)n_units, &d_units))) return rc;
    if ((rc = stage_in(c, c->st[18], global_lens, (size_t)n_global_total, &d_gl))) return rc;
    if ((rc = stage_in(c, c->st[19], target_lens, (size_t)n_target_total, &d_tl))) return rc;
    if ((rc = stage_out(c, c->st[20], (size_t)n_units, &d_calls))) return rc;
    if ((rc = stage_out(c, c->st[21], marg_n, &d_marg))) return rc;
    if ((rc = stage_in(c, c->st[22], joint_off, (size_t)n_units + 1, &d_joff))) return rc;
    if ((rc = stage_out(c, c->st[23], joint_len, &d_joint))) return rc;
    if ((rc = stage_out(c, c->st[24], (size_t)n_units, &d_jn))) return rc;
    if ((rc = stage_out(c, c->st[25], (size_t)n_units, &d_jt))) return rc;
    if ((rc = stage_in(c, c->st[26], (const int32_t*)unit_task.data(), (size_t)n_units, &d_task))) return rc;
    if ((rc = stage_in(c, c->st[27], unit_word_off, (size_t)n_units + 1, &d_uw))) return rc;
    if ((rc = stage_in(c, c->st[28], unit_seq4_off, (size_t)n_units + 1, &d_us))) return rc;
    if ((rc = stage_in(c, c->st[29], unit_name_off, (size_t)n_units + 1, &d_un))) return rc;
    if ((rc = stage_out(c, c->st[30], s4_bytes, &d_seq4))) return rc;
    if ((rc = stage_out(c, c->st[31], (size_t)n_reads + 1, &d_s4off))) return rc;
    if ((rc = stage_out(c, c->st[32], nm_bytes, &d_names))) return rc;
    if ((rc = stage_out(c, c->st[33], (size_t)n_reads + 1, &d_nmoff))) return rc;
    g = 0;
    for (int32_t sgi = 0; sgi < n_segs; ++sgi) {
        HIPCHK(c, tredgpu_front::launch_pack_selected(views[sgi].out, views[sgi].sel_list, d_task, d_uoff, d_uw, d_us, d_un, g, segs[sgi].n_units, d_packed,
                                                      d_off, d_len, d_seq4, d_s4off, d_names, d_nmoff, c->stream));
        g += segs[sgi].n_units;
    }
    if (n_reads > 0 && (rc = run_sw_device(c, d_packed, d_off, d_len, n_reads, d_uoff, d_ulad, n_units, params, max_len, d_tag, d_h, d_score, nullptr, 0)))
        return rc;
    {
        ScopedTimer tm(c, TREDGPU_KERNEL_TALLY);
        HIPCHK(c, launch_tally(d_tag, d_h, n_reads, d_uoff, n_units, nullptr, hist_stride, d_f, d_p, d_r, (uint8_t*)c->ws_drop.p, c->stream));
    }
    if ((rc = run_grid_device(c, d_units, n_units, hist_stride, d_f, d_p, d_r, d_gl, d_tl, d_calls, nullptr, nullptr, d_marg, marg_stride, limits, d_joff,
                              d_joint, d_jn, d_jt)))
        return rc;
    if ((rc = copy_back(c, out_tag, (const uint8_t*)d_tag, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_h, (const int16_t*)d_h, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, out_score, (const int16_t*)d_score, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, rept_cnt, (const int32_t*)d_r, rept_cnt ? hn : 0))) return rc;
    if ((rc = copy_back(c, joint, (const double*)d_joint, joint_len))) return rc;
    if ((rc = copy_back(c, joint_n, (const int32_t*)d_jn, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, joint_total, (const double*)d_jt, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, calls, (const tredgpu_call*)d_calls, (size_t)n_units))) return rc;
    if ((rc = copy_back(c, marg, (const double*)d_marg, marg_n))) return rc;
    if ((rc = copy_back(c, read_len, (const int32_t*)d_len, (size_t)n_reads))) return rc;
    if ((rc = copy_back(c, seq4_off, (const int64_t*)d_s4off, n_reads > 0 ? (size_t)n_reads + 1 : 0))) return rc;
    if ((rc = copy_back(c, seq4, (const uint8_t*)d_seq4, s4_bytes))) return rc;
    if ((rc = copy_back(c, name_off, (const int64_t*)d_nmoff, n_reads > 0 ? (size_t)n_reads + 1 : 0))) return rc;
    if ((rc = copy_back(c, (uint8_t*)names, (const uint8_t*)d_names, nm_bytes))) return rc;
    HIPCHK(c, stream_sync(c));
    return 0;
}

int tredgpu_genotype_batch(tredgpu_ctx* c, int mem, const uint32_t* packed, const int64_t* read_off,
                           const int32_t* read_len, int64_t n_reads, const int32_t* unit_read_off,
                           const int32_t* unit_ladder, const tredgpu_unit_params* units, int32_t n_units,
                           const tredgpu_sw_params* params, const int32_t* read_pair_id,
                           const int32_t* global_lens, int64_t n_global_total, const int32_t* target_lens,
                           int64_t n_target_total, uint8_t* out_tag, int16_t* out_h, int16_t* out_score,
                           int32_t hist_stride, int32_t* full_cnt, int32_t* pref_cnt, int32_t* rept_cnt,
                           tredgpu_call* calls) {
    if (!c) return -2;
    int rc;
    if ((rc = check_sw_params(c, params))) return rc;
    if ((rc = check_grid_common(c, units, n_units))) return rc;
    if (c->h_ladders.empty()) return fail(c, -4, "no ladders registered (tredgpu_set_ladders)");
    if (hist_stride <= c->max_ladder_units) return fail(c, -2, "hist_stride %d must exceed the largest max_units %d", hist_stride, c->max_ladder_units);
    if (n_reads < 0) return fail(c, -2, "negative n_reads");
    if (n_units > 0 && (!unit_read_off || !unit_ladder || !full_cnt || !pref_cnt || !rept_cnt || !calls)) return fail(c, -2, "NULL array argument");
    if (n_reads > 0 && (!packed || !read_off || !read_len || !out_tag || !out_h || !out_score)) return fail(c, -2, "NULL array argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (mem == TREDGPU_MEM_DEVICE) {
        int max_len = params->max_read_len > 0 ? params->max_read_len : TREDGPU_ASSUMED_READ_LEN;
        int limits[2] = {0, 0};
        if (n_units > 0 && (rc = query_max_insert(c, units, n_units, limits))) return rc;
        if (read_pair_id && (rc = ensure(c, c->ws_drop, (size_t)n_reads))) return rc;
        if ((rc = run_sw_device(c, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, n_units, params,
                                max_len, out_tag, out_h, out_score, nullptr, 0)))
            return rc;
        {
            ScopedTimer tm(c, TREDGPU_KERNEL_TALLY);
            HIPCHK(c, launch_tally(out_tag, out_h, n_reads, unit_read_off, n_units, read_pair_id, hist_stride, full_cnt,
                                   pref_cnt, rept_cnt, (uint8_t*)c->ws_drop.p, c->stream));
        }
        return run_grid_device(c, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens, target_lens,
                               calls, nullptr, nullptr, nullptr, 0, limits);
    }
    if (mem != TREDGPU_MEM_HOST) return fail(c, -2, "bad mem");
    // HOST memory: compose the three host-memory calls (each validates and stages its own arguments)
    if ((rc = tredgpu_sw_classify(c, mem, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, n_units,
                                  params, out_tag, out_h, out_score, nullptr, 0)))
        return rc;
    if ((rc = tredgpu_tally(c, mem, out_tag, out_h, n_reads, unit_read_off, n_units, read_pair_id, hist_stride,
                            full_cnt, pref_cnt, rept_cnt)))
        return rc;
    return tredgpu_likelihood_grid(c, mem, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens,
                                   n_global_total, target_lens, n_target_total, calls, nullptr, nullptr, nullptr, 0);
}

}  // extern "C"
