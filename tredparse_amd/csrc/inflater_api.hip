// inflater_api.hip -- the C ABI of the front end on the device (include/tredgpu.h section 4: tredgpu_inflater_*,
// tredgpu_inflate_blocks[_crc], tredgpu_inflate_walk, tredgpu_inflater_fetch[_dense]): buffers, streams, the order of the
// launches.  The kernels are inflate_decode.hip's and walk.hip's (inflater_internal.h).
#include <hip/hip_runtime.h>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <cstdio>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "inflater_internal.h"

using namespace tredgpu_front;

// ---- C ABI (include/tredgpu.h) ------------------------------------------------------------------------------------
// A call is cut into slices of blocks that alternate between two streams: the copy-in and the decoding of slice k + 1
// run beside the copy-out of slice k (the copy-out is the long pole: four bytes leave for every byte that arrives).
constexpr int MAX_SLICES = 8;
constexpr int SLICE_BLOCKS = 4096;      // >= 4 096 wavefronts per launch: 16 per CU, and two launches run side by side

struct tredgpu_inflater {
    int device = 0;
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};   // waited for asleep (polled): see tredgpu_inflate_blocks
    hipEvent_t t0[2] = {}, t1[2] = {}, k0[MAX_SLICES] = {}, k1[MAX_SLICES] = {};   // timing (tredgpu_inflater_timing)
    int last_slices = 0, last_streams = 0;
    uint8_t *h_comp = nullptr, *h_out = nullptr;      // pinned staging the caller fills / reads in place
    bool host_out = true;                             // false (tredgpu_inflater_host_out): no pinned room for the whole output --
                                                      // the blocks the host wants come through tredgpu_inflater_fetch_dense
    uint8_t* h_dense = nullptr; size_t cap_dense = 0; // pinned: the fetched blocks, one after the other
    uint8_t *h_pieces = nullptr, *d_pieces = nullptr; size_t cap_pieces = 0;   // the fetch kernel's copy table (FetchPiece)
    int64_t *h_off = nullptr;                         // pinned: comp_off[n+1] then out_off[n+1]
    int32_t* h_status = nullptr;                      // pinned: status[n] then crc[n]
    size_t cap_comp = 0, cap_out = 0, cap_blocks = 0;
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    int64_t* d_off = nullptr;
    int32_t* d_status = nullptr;
    // the pair walk (tredgpu_inflate_walk): a stream of its own, the file's view of the blocks, tasks, per-task tables, pools
    hipStream_t wstream = nullptr, astream = nullptr;          // (astream: the alternative loci's walks, beside the pair walks)
    hipEvent_t adone = nullptr;
    hipEvent_t wdone = nullptr, w0 = nullptr, w1 = nullptr, decoded[2] = {nullptr, nullptr};
    bool walk_timed = false, big_lds_allowed = false;
    int walk_table_cap = 0;
    size_t last_walk_tasks = 0;                       // regions of the last walk call (tredgpu_inflater_walk_serial_regions)
    uint8_t* d_wblk = nullptr;  size_t cap_wblk = 0;        // bcoff[n] int64, then bclen[n] int32, then xcrc[n] uint32
    uint8_t* h_wblk = nullptr;                               // pinned, same layout
    uint8_t* d_wtask = nullptr; uint8_t* h_wtask = nullptr; size_t cap_wtask = 0;   // tasks then chunks
    uint8_t* d_wres = nullptr;  uint8_t* h_wres = nullptr;  size_t cap_wres = 0;    // results then the two counters
    WalkPair* d_wpairs = nullptr; size_t cap_wscratch = 0;   // in tasks
    WalkRec* d_wrecs = nullptr; WalkFields* d_wfields = nullptr; size_t cap_wrecs = 0;          // in records: the chain's list, the parsed fields
    WalkChained* d_wchained = nullptr; size_t cap_wchained = 0;
    uint8_t* d_atask = nullptr; uint8_t* h_atask = nullptr; size_t cap_atask = 0;   // the alternative loci's tasks then chunks
    uint8_t* d_ares = nullptr;  uint8_t* h_ares = nullptr;  size_t cap_ares = 0;    // their results, then the blocks' need flags
    // the read selection (tredgpu.h section 5): per task its parameters and result, and TREDGPU_SELECT_CAP record places
    uint8_t* d_sel = nullptr; uint8_t* h_sel = nullptr; size_t cap_sel = 0;        // tasks' tredgpu_select_task, then their results
    int64_t* d_sel_list = nullptr; size_t cap_sel_list = 0;                        // in tasks
    int sel_tasks = 0;                                                             // tasks of the last walk with a selection (0: none)
    hipEvent_t aready = nullptr;                                                   // the alternative loci's walk has run (the selection reads its hits)
    int32_t *d_gpool = nullptr, *d_tpool = nullptr, *h_gpool = nullptr, *h_tpool = nullptr;
    size_t cap_gpool = 0, cap_tpool = 0;          // (device pools: the call's bound)
    size_t cap_hgpool = 0, cap_htpool = 0;        // (pinned host pools: what the walks really produced, an eighth more)
    std::string err;
};

namespace {
thread_local std::string g_inflate_error;

// TREDGPU_TRACE=1: host-side timestamps of a call's phases on stderr (milliseconds since the call began)
struct CallTrace {
    bool on; const char* what; double t0; std::string line;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    explicit CallTrace(const char* w) : on(getenv("TREDGPU_TRACE") != nullptr), what(w), t0(on ? now() : 0) {}
    void mark(const char* k) { if (on) { char b[64]; snprintf(b, sizeof b, " %s=%.2f", k, now() - t0); line += b; } }
    ~CallTrace() { if (on) fprintf(stderr, "[tredgpu %d] %s:%s\n", (int)getpid(), what, line.c_str()); }
};


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
    if (f->h_dense) (void)hipHostFree(f->h_dense);
    f->h_dense = nullptr; f->cap_dense = 0;
    if (f->h_pieces) (void)hipHostFree(f->h_pieces);
    if (f->d_pieces) (void)hipFree(f->d_pieces);
    f->h_pieces = f->d_pieces = nullptr; f->cap_pieces = 0;
    if (f->h_off) (void)hipHostFree(f->h_off);
    if (f->h_status) (void)hipHostFree(f->h_status);
    for (void* p : {(void*)f->d_comp, (void*)f->d_out, (void*)f->d_off, (void*)f->d_status})
        if (p) (void)hipFree(p);
    f->h_comp = f->h_out = nullptr; f->h_off = nullptr; f->h_status = nullptr;
    f->d_comp = f->d_out = nullptr; f->d_off = nullptr; f->d_status = nullptr;
    f->cap_comp = f->cap_out = f->cap_blocks = 0;
}

void release_walk(tredgpu_inflater* f) {
    for (void* p : {(void*)f->h_wblk, (void*)f->h_wtask, (void*)f->h_wres, (void*)f->h_gpool, (void*)f->h_tpool, (void*)f->h_atask, (void*)f->h_ares, (void*)f->h_sel})
        if (p) (void)hipHostFree(p);
    for (void* p : {(void*)f->d_sel, (void*)f->d_sel_list})
        if (p) (void)hipFree(p);
    f->h_sel = f->d_sel = nullptr; f->d_sel_list = nullptr; f->cap_sel = f->cap_sel_list = 0; f->sel_tasks = 0;
    for (void* p : {(void*)f->d_wblk, (void*)f->d_wtask, (void*)f->d_wres, (void*)f->d_wpairs, (void*)f->d_gpool, (void*)f->d_tpool, (void*)f->d_atask, (void*)f->d_ares,
                    (void*)f->d_wrecs, (void*)f->d_wfields, (void*)f->d_wchained})
        if (p) (void)hipFree(p);
    f->h_wblk = f->h_wtask = f->h_wres = f->h_atask = f->h_ares = nullptr; f->h_gpool = f->h_tpool = nullptr;
    f->d_wblk = f->d_wtask = f->d_wres = f->d_atask = f->d_ares = nullptr; f->d_wpairs = nullptr; f->d_gpool = f->d_tpool = nullptr;
    f->d_wrecs = nullptr; f->d_wfields = nullptr; f->d_wchained = nullptr; f->cap_wrecs = f->cap_wchained = 0;
    f->cap_wblk = f->cap_wtask = f->cap_wres = f->cap_wscratch = f->cap_gpool = f->cap_tpool = f->cap_atask = f->cap_ares = 0;
    f->cap_hgpool = f->cap_htpool = 0;
}

// grow-only pairs of pinned host / device buffers for the walk's small arrays
int grow_pair(tredgpu_inflater* f, uint8_t** host, uint8_t** dev, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    const size_t c = std::max(need + need / 8, *cap + *cap / 2);     // (an eighth past the call that makes it grow: see tredgpu_inflater_reserve)
    if (*host) (void)hipHostFree(*host);
    if (*dev) (void)hipFree(*dev);
    *host = nullptr; *dev = nullptr; *cap = 0;
    ICHK(f, hipHostMalloc((void**)host, c, hipHostMallocDefault));
    ICHK(f, hipMalloc((void**)dev, c));
    *cap = c;
    return 0;
}

void destroy_handles(tredgpu_inflater* f) {
    for (hipEvent_t e : {f->done[0], f->done[1], f->t0[0], f->t0[1], f->t1[0], f->t1[1]}) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < MAX_SLICES; ++k) { if (f->k0[k]) (void)hipEventDestroy(f->k0[k]); if (f->k1[k]) (void)hipEventDestroy(f->k1[k]); }
    for (hipEvent_t e : {f->wdone, f->w0, f->w1, f->decoded[0], f->decoded[1]}) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : f->stream) if (st) (void)hipStreamDestroy(st);
    if (f->wstream) (void)hipStreamDestroy(f->wstream);
    if (f->astream) (void)hipStreamDestroy(f->astream);
    if (f->adone) (void)hipEventDestroy(f->adone);
    if (f->aready) (void)hipEventDestroy(f->aready);
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
    // several of these
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    e = hipSetDevice(device_id);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = create_partitioned_stream(&f->stream[k], hipStreamNonBlocking, lo_prio, 0, device_id);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->done[k], hipEventBlockingSync | hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&f->t0[k]);
        if (e == hipSuccess) e = hipEventCreate(&f->t1[k]);
    }
    for (int k = 0; k < MAX_SLICES && e == hipSuccess; ++k) {
        e = hipEventCreate(&f->k0[k]);
        if (e == hipSuccess) e = hipEventCreate(&f->k1[k]);
    }
    if (e == hipSuccess) e = create_partitioned_stream(&f->wstream, hipStreamNonBlocking, lo_prio, 0, device_id);
    if (e == hipSuccess) e = create_partitioned_stream(&f->astream, hipStreamNonBlocking, lo_prio, 0, device_id);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->adone, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->aready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->wdone, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&f->w0);
    if (e == hipSuccess) e = hipEventCreate(&f->w1);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&f->decoded[k], hipEventDisableTiming);
    if (e != hipSuccess) {
        destroy_handles(f);
        delete f;
        return ifail(nullptr, -10, "stream / event creation", e);
    }
    *out = f;
    return 0;
}

void tredgpu_inflater_destroy(tredgpu_inflater* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (hipStream_t st : f->stream) (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(f->wstream);
    if (f->astream) (void)hipStreamSynchronize(f->astream);
    release(f);
    release_walk(f);
    destroy_handles(f);
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
        for (hipStream_t st : f->stream) ICHK(f, hipStreamSynchronize(st));
        // (page-locked staging grows by an eighth past the largest call seen, and a sixteenth past the call that makes it grow:
        //  a chunk's size varies by a few per cent -- sized exactly, the second and the third chunk of a driver each freed and
        //  page-locked 0.3 GB again, 0.3-0.4 s of a cohort's first two seconds --, and every spare byte here is pinned three times
        //  per driver process)
        const size_t cc = std::max(need_c + need_c / 16, f->cap_comp + f->cap_comp / 8), co = std::max(need_o + need_o / 16, f->cap_out + f->cap_out / 8),
                     cb = std::max(need_b + need_b / 16, f->cap_blocks + f->cap_blocks / 2);
        release(f);
        ICHK(f, hipHostMalloc((void**)&f->h_comp, cc, hipHostMallocDefault));
        if (f->host_out) ICHK(f, hipHostMalloc((void**)&f->h_out, co, hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_off, 2 * cb * sizeof(int64_t), hipHostMallocDefault));
        ICHK(f, hipHostMalloc((void**)&f->h_status, 2 * cb * sizeof(int32_t), hipHostMallocDefault));
        ICHK(f, hipMalloc((void**)&f->d_comp, cc));
        ICHK(f, hipMalloc((void**)&f->d_out, co));
        ICHK(f, hipMalloc((void**)&f->d_off, 2 * cb * sizeof(int64_t)));
        ICHK(f, hipMalloc((void**)&f->d_status, 2 * cb * sizeof(int32_t)));
        f->cap_comp = cc; f->cap_out = co; f->cap_blocks = cb;
    }
    *comp_host = f->h_comp;
    *out_host = f->h_out;
    *comp_off_host = f->h_off;
    *out_off_host = f->h_off + f->cap_blocks;
    return 0;
}

}  // extern "C"

namespace {
int wait_asleep(tredgpu_inflater* f, hipEvent_t ev) {
    // hipEventSynchronize spins even on a hipEventBlockingSync event here (measured: CPU time = wall time, and calls of
    // other threads on other streams queue up behind the spinning one); the host threads that wait are the ones whose
    // cores the path is short of
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return ifail(f, -10, "hipEventQuery", q);
        usleep(100);
    }
}

// copy in, decode (in slices on two streams); copy_out: every slice's blocks go back as soon as they are decoded.
// w != nullptr: the pair walk follows the last slice on a stream of its own, and its results come back.
int run_inflate(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc, bool copy_out, tredgpu_walk_args* w) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || (n_blocks > 0 && !status)) return ifail(f, -2, "bad arguments (reserve first)");
    if (n_blocks == 0) return 0;
    if (copy_out && !f->h_out) return ifail(f, -2, "this inflater keeps no host copy of the output (tredgpu_inflater_host_out): walk and fetch");
    CallTrace tr(w ? "inflate_walk" : "inflate");
    const int64_t* coff = f->h_off;
    const int64_t* ooff = f->h_off + f->cap_blocks;
    for (int32_t k = 0; k < n_blocks; ++k) {
        if (coff[k] < 0 || (coff[k] & 3) != 0 || coff[k + 1] < coff[k] || ooff[k] < 0 || ooff[k + 1] < ooff[k] ||
            ooff[k + 1] - ooff[k] > 65536)
            return ifail(f, -2, "block offsets: payloads start on 4-byte boundaries, ascend, and inflate to at most 64 KiB each");
    }
    if ((size_t)coff[n_blocks] + 64 > f->cap_comp || (size_t)ooff[n_blocks] + 64 > f->cap_out) return ifail(f, -2, "offsets beyond the reserved buffers");
    ICHK(f, hipSetDevice(f->device));
    const int slices = std::min(MAX_SLICES, std::max(1, n_blocks / SLICE_BLOCKS));
    const int nstreams = slices > 1 ? 2 : 1;
    int64_t* d_coff = f->d_off;
    int64_t* d_ooff = f->d_off + f->cap_blocks;
    int32_t* d_crc = f->d_status + f->cap_blocks;
    const bool want_crc = crc != nullptr || w != nullptr;
    // ---- the walk's inputs go first, on its own stream (nothing there depends on the decoding yet) ----
    size_t n_tasks = 0, n_chunks = 0, n_alt = 0, n_alt_chunks = 0, total_recs = 0;
    if (w) {
        if (w->n_tasks < 0 || w->n_chunks < 0 || !w->blk_coffset || !w->blk_clen || !w->blk_crc || (w->n_tasks > 0 && (!w->tasks || !w->results)) ||
            (w->n_chunks > 0 && !w->chunks) || w->cap_global < 0 || w->cap_target < 0 || (w->cap_global > 0 && !w->global_pool) ||
            (w->cap_target > 0 && !w->target_pool))
            return ifail(f, -2, "bad walk arguments");
        n_tasks = (size_t)w->n_tasks; n_chunks = (size_t)w->n_chunks;
        for (size_t t = 0; t < n_tasks; ++t) {
            const tredgpu_walk_task& T = w->tasks[t];
            if (T.n_chunks >= 0 && (T.chunk_first < 0 || (size_t)T.chunk_first + (size_t)T.n_chunks > n_chunks || T.block_first < 0 ||
                                    T.block_end > n_blocks || T.block_first > T.block_end))
                return ifail(f, -2, "walk task outside its chunks / blocks");
        }
        for (size_t q = 0; q < n_chunks; ++q)
            if (w->chunks[q].begin_upos < 0 || w->chunks[q].begin_upos > 65536) return ifail(f, -2, "walk chunk starts outside a block");
        n_alt = (size_t)std::max(w->n_alt_tasks, 0);
        n_alt_chunks = (size_t)std::max(w->n_alt_chunks, 0);
        if (w->n_alt_tasks < 0 || w->n_alt_chunks < 0 || (n_alt > 0 && (!w->alt_tasks || !w->alt_results || !w->need)) || (n_alt_chunks > 0 && !w->alt_chunks))
            return ifail(f, -2, "bad walk arguments (alternative loci)");
        for (size_t t = 0; t < n_alt; ++t) {
            const tredgpu_walk_task& T = w->alt_tasks[t];
            if (T.n_chunks >= 0 && (T.chunk_first < 0 || (size_t)T.chunk_first + (size_t)T.n_chunks > n_alt_chunks || T.block_first < 0 ||
                                    T.block_end > n_blocks || T.block_first > T.block_end))
                return ifail(f, -2, "walk task outside its chunks / blocks");
        }
        for (size_t q = 0; q < n_alt_chunks; ++q)
            if (w->alt_chunks[q].begin_upos < 0 || w->alt_chunks[q].begin_upos > 65536) return ifail(f, -2, "walk chunk starts outside a block");
        if ((w->select != nullptr) != (w->selected != nullptr)) return ifail(f, -2, "select and selected go together");
        if (w->select) {
            for (size_t t = 0; t < n_tasks; ++t) {
                const tredgpu_select_task& S = w->select[t];
                if (S.n_alt > 0 && (S.alt_first < 0 || (size_t)S.alt_first + (size_t)S.n_alt > n_alt)) return ifail(f, -2, "select task outside the alternative loci's tasks");
            }
            if (grow_pair(f, &f->h_sel, &f->d_sel, &f->cap_sel, n_tasks * (sizeof(tredgpu_select_task) + sizeof(tredgpu_select_result)) + 64)) return -10;
            if (n_tasks > f->cap_sel_list) {
                const size_t c = std::max(n_tasks, f->cap_sel_list + f->cap_sel_list / 2);
                if (f->d_sel_list) (void)hipFree(f->d_sel_list);
                f->d_sel_list = nullptr; f->cap_sel_list = 0;
                ICHK(f, hipMalloc((void**)&f->d_sel_list, c * TREDGPU_SELECT_CAP * sizeof(int64_t)));
                f->cap_sel_list = c;
            }
            memcpy(f->h_sel, w->select, n_tasks * sizeof(tredgpu_select_task));
        }
        f->sel_tasks = 0;
        const size_t nb = (size_t)n_blocks;
        if (n_alt > 0) {
            if (grow_pair(f, &f->h_atask, &f->d_atask, &f->cap_atask, n_alt * sizeof(tredgpu_walk_task) + n_alt_chunks * sizeof(tredgpu_walk_chunk) + 64)) return -10;
            if (grow_pair(f, &f->h_ares, &f->d_ares, &f->cap_ares, n_alt * sizeof(tredgpu_alt_result) + nb + 64)) return -10;
            memcpy(f->h_atask, w->alt_tasks, n_alt * sizeof(tredgpu_walk_task));
            memcpy(f->h_atask + n_alt * sizeof(tredgpu_walk_task), w->alt_chunks, n_alt_chunks * sizeof(tredgpu_walk_chunk));
            ICHK(f, hipMemcpyAsync(f->d_atask, f->h_atask, n_alt * sizeof(tredgpu_walk_task) + n_alt_chunks * sizeof(tredgpu_walk_chunk), hipMemcpyHostToDevice, f->astream));
            ICHK(f, hipMemsetAsync(f->d_ares + n_alt * sizeof(tredgpu_alt_result), 0, nb, f->astream));
        }
        if (grow_pair(f, &f->h_wblk, &f->d_wblk, &f->cap_wblk, nb * 16 + 64)) return -10;
        // tasks | chunks | rec_base[n_tasks + 1] (8-byte entries behind 8-byte-sized structs: aligned)
        const size_t rb_at = n_tasks * sizeof(tredgpu_walk_task) + n_chunks * sizeof(tredgpu_walk_chunk);
        static_assert(sizeof(tredgpu_walk_task) % 8 == 0 && sizeof(tredgpu_walk_chunk) % 8 == 0, "rec_base stays 8-byte aligned");
        if (grow_pair(f, &f->h_wtask, &f->d_wtask, &f->cap_wtask, rb_at + (n_tasks + 1) * sizeof(int64_t) + 64)) return -10;
        // room for every region's record list: the blocks its chunks span (a task's block_first .. block_end is its whole
        // FILE), at no less than 64 bytes per record (36 fixed bytes, a name, 36 bases and their qualities: 100 and more) --
        // a region that has more records than that is the host's (status 4)
        int64_t* rec_base = (int64_t*)(f->h_wtask + rb_at);
        rec_base[0] = 0;
        for (size_t t = 0; t < n_tasks; ++t) {
            const tredgpu_walk_task& T = w->tasks[t];
            int64_t bytes = 0;
            for (int32_t q = 0; q < T.n_chunks; ++q) {
                const tredgpu_walk_chunk& ch = w->chunks[T.chunk_first + q];
                if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) continue;
                const int64_t* lo = w->blk_coffset + ch.begin_block;
                const int64_t* hi = std::upper_bound(lo, w->blk_coffset + T.block_end, (int64_t)(ch.end_voffset >> 16));
                bytes += ooff[ch.begin_block + (hi - lo)] - ooff[ch.begin_block];
            }
            rec_base[t + 1] = rec_base[t] + (T.n_chunks > 0 ? bytes / 64 + 64 : 0);
        }
        total_recs = (size_t)rec_base[n_tasks];
        if (total_recs > f->cap_wrecs) {
            const size_t c = std::max(total_recs + total_recs / 8, f->cap_wrecs + f->cap_wrecs / 2);
            if (f->d_wrecs) (void)hipFree(f->d_wrecs);
            if (f->d_wfields) (void)hipFree(f->d_wfields);
            f->d_wrecs = nullptr; f->d_wfields = nullptr; f->cap_wrecs = 0;
            ICHK(f, hipMalloc((void**)&f->d_wrecs, c * sizeof(WalkRec)));
            ICHK(f, hipMalloc((void**)&f->d_wfields, c * sizeof(WalkFields)));
            f->cap_wrecs = c;
        }
        if (n_tasks > f->cap_wchained) {
            const size_t c = std::max(n_tasks + n_tasks / 8, f->cap_wchained + f->cap_wchained / 2);
            if (f->d_wchained) (void)hipFree(f->d_wchained);
            f->d_wchained = nullptr; f->cap_wchained = 0;
            ICHK(f, hipMalloc((void**)&f->d_wchained, c * sizeof(WalkChained)));
            f->cap_wchained = c;
        }
        if (grow_pair(f, &f->h_wres, &f->d_wres, &f->cap_wres, n_tasks * sizeof(tredgpu_walk_result) + 64)) return -10;
        if (n_tasks > f->cap_wscratch) {
            const size_t c = std::max(n_tasks + n_tasks / 8, f->cap_wscratch + f->cap_wscratch / 2);
            if (f->d_wpairs) (void)hipFree(f->d_wpairs);
            f->d_wpairs = nullptr; f->cap_wscratch = 0;
            ICHK(f, hipMalloc((void**)&f->d_wpairs, c * WALK_PAIR_CAP * sizeof(WalkPair)));
            f->cap_wscratch = c;
        }
        // (the pools on the device hold the call's BOUND -- a tenth of it is used at 30x --, their pinned host copies are
        //  sized behind the walks, from what was really produced)
        for (int which = 0; which < 2; ++which) {
            int32_t** d = which ? &f->d_tpool : &f->d_gpool;
            size_t* cap = which ? &f->cap_tpool : &f->cap_gpool;
            const size_t need = ((size_t)(which ? w->cap_target : w->cap_global) + 16) * 4;
            if (need > *cap) {
                const size_t c = std::max(need + need / 8, *cap + *cap / 2);
                if (*d) (void)hipFree(*d);
                *d = nullptr; *cap = 0;
                ICHK(f, hipMalloc((void**)d, c));
                *cap = c;
            }
        }
        memcpy(f->h_wblk, w->blk_coffset, nb * 8);
        memcpy(f->h_wblk + nb * 8, w->blk_clen, nb * 4);
        memcpy(f->h_wblk + nb * 12, w->blk_crc, nb * 4);
        memcpy(f->h_wtask, w->tasks, n_tasks * sizeof(tredgpu_walk_task));
        memcpy(f->h_wtask + n_tasks * sizeof(tredgpu_walk_task), w->chunks, n_chunks * sizeof(tredgpu_walk_chunk));
        ICHK(f, hipMemcpyAsync(f->d_wblk, f->h_wblk, nb * 16, hipMemcpyHostToDevice, f->wstream));
        ICHK(f, hipMemcpyAsync(f->d_wtask, f->h_wtask, rb_at + (n_tasks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->wstream));
        ICHK(f, hipMemsetAsync(f->d_wres + n_tasks * sizeof(tredgpu_walk_result), 0, 16, f->wstream));
        if (w->select && n_tasks > 0) ICHK(f, hipMemcpyAsync(f->d_sel, f->h_sel, n_tasks * sizeof(tredgpu_select_task), hipMemcpyHostToDevice, f->wstream));
    }
    tr.mark("walk_inputs");
    // the two streams never wait for each other: each copies the offsets in for itself (both write the same values)
    for (int s = 0; s < nstreams; ++s) {
        ICHK(f, hipEventRecord(f->t0[s], f->stream[s]));
        ICHK(f, hipMemcpyAsync(d_coff, coff, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream[s]));
        ICHK(f, hipMemcpyAsync(d_ooff, ooff, ((size_t)n_blocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, f->stream[s]));
    }
    for (int k = 0; k < slices; ++k) {
        hipStream_t st = f->stream[k % nstreams];
        const int32_t b0 = (int32_t)((int64_t)n_blocks * k / slices), b1 = (int32_t)((int64_t)n_blocks * (k + 1) / slices);
        const size_t c_from = (size_t)coff[b0], c_to = std::min(((size_t)coff[b1] + 3) & ~(size_t)3, f->cap_comp);
        ICHK(f, hipMemcpyAsync(f->d_comp + c_from, f->h_comp + c_from, c_to - c_from, hipMemcpyHostToDevice, st));
        ICHK(f, hipEventRecord(f->k0[k], st));
        ICHK(f, launch_inflate((const uint32_t*)f->d_comp, d_coff, f->d_out, d_ooff, b0, b1 - b0, f->d_status, want_crc ? (uint32_t*)d_crc : nullptr, st));
        ICHK(f, hipEventRecord(f->k1[k], st));
        if (copy_out) ICHK(f, hipMemcpyAsync(f->h_out + ooff[b0], f->d_out + ooff[b0], (size_t)(ooff[b1] - ooff[b0]), hipMemcpyDeviceToHost, st));
        ICHK(f, hipMemcpyAsync(f->h_status + b0, f->d_status + b0, (size_t)(b1 - b0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (want_crc) ICHK(f, hipMemcpyAsync(f->h_status + f->cap_blocks + b0, d_crc + b0, (size_t)(b1 - b0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    f->last_slices = slices;
    f->last_streams = nstreams;
    f->walk_timed = false;
    if (!w) f->sel_tasks = 0;                      // (the output a selection pointed into is being overwritten)
    if (w) {
        // the walk reads every slice's blocks: its stream waits for the last launch of both decode streams
        for (int s = 0; s < nstreams; ++s) {
            ICHK(f, hipEventRecord(f->decoded[s], f->stream[s]));
            ICHK(f, hipStreamWaitEvent(f->wstream, f->decoded[s], 0));
            ICHK(f, hipStreamWaitEvent(f->astream, f->decoded[s], 0));
        }
        const size_t nb = (size_t)n_blocks;
        WalkView v;
        v.out = f->d_out; v.ooff = d_ooff; v.bstatus = f->d_status; v.bcrc = (const uint32_t*)d_crc;
        v.bcoff = (const int64_t*)f->d_wblk; v.bclen = (const int32_t*)(f->d_wblk + nb * 8); v.xcrc = (const uint32_t*)(f->d_wblk + nb * 12);
        v.out_end = ooff[n_blocks] + 48;                   // (the buffers hold 64 bytes more than reserved)
        ICHK(f, hipEventRecord(f->w0, f->wstream));
        if (n_tasks > 0) {
            // the small table when every region is short: a region of up to 40 blocks (2.6 MB of records, ~8 000 of them)
            // has at most ~4 000 names; one that has more after all is handed back to the host (status 4)
            int table_cap = WALK_PAIR_CAP_SMALL;
            for (size_t t = 0; t < n_tasks && table_cap == WALK_PAIR_CAP_SMALL; ++t) {
                const tredgpu_walk_task& T = w->tasks[t];
                int64_t span = 0;
                for (int32_t q = 0; q < T.n_chunks; ++q) {
                    const tredgpu_walk_chunk& ch = w->chunks[T.chunk_first + q];
                    if (ch.begin_block < T.block_first || ch.begin_block >= T.block_end) continue;
                    const int64_t* lo = w->blk_coffset + ch.begin_block;
                    const int64_t* hi = std::upper_bound(lo, w->blk_coffset + T.block_end, (int64_t)(ch.end_voffset >> 16));
                    span += hi - lo;
                }
                if (span > 40) table_cap = WALK_PAIR_CAP;
            }
            if (!f->big_lds_allowed) {                 // (per inflater: each lives on one device, and a process may use several)
                ICHK(f, allow_pair_walk_lds());
                f->big_lds_allowed = true;
            }
            f->walk_table_cap = table_cap;
            f->last_walk_tasks = n_tasks;
            const tredgpu_walk_task* d_tasks = (const tredgpu_walk_task*)f->d_wtask;
            const tredgpu_walk_chunk* d_chunks = (const tredgpu_walk_chunk*)(f->d_wtask + n_tasks * sizeof(tredgpu_walk_task));
            const int64_t* d_rec_base = (const int64_t*)(f->d_wtask + n_tasks * sizeof(tredgpu_walk_task) + n_chunks * sizeof(tredgpu_walk_chunk));
            ICHK(f, launch_walk_chain_par(v, d_tasks, d_chunks, d_rec_base, f->d_wrecs, f->d_wchained, (int)n_tasks, f->wstream));
            if (getenv("TREDGPU_WALK_SERIAL") != nullptr)       // (A/B and tests: every region through the serial chain)
                ICHK(f, hipMemsetAsync(f->d_wchained, 0, n_tasks * sizeof(WalkChained), f->wstream));
            ICHK(f, launch_walk_chain(v, d_tasks, d_chunks, d_rec_base, f->d_wrecs, f->d_wchained, (int)n_tasks, f->wstream));
            ICHK(f, launch_walk_parse(v, (int)n_tasks, d_rec_base, f->d_wrecs, f->d_wchained, f->d_wfields, total_recs, f->wstream));
            ICHK(f, launch_pair_walk(v, d_tasks, d_rec_base, f->d_wrecs, f->d_wfields, f->d_wchained, (tredgpu_walk_result*)f->d_wres, f->d_wpairs,
                                     f->d_gpool, w->cap_global, f->d_tpool, w->cap_target,
                                     (unsigned long long*)(f->d_wres + n_tasks * sizeof(tredgpu_walk_result)), table_cap, (int)n_tasks, f->wstream));
        }
        if (!(w->select && n_tasks > 0)) ICHK(f, hipEventRecord(f->w1, f->wstream));
        f->walk_timed = true;
        ICHK(f, hipMemcpyAsync(f->h_wres, f->d_wres, n_tasks * sizeof(tredgpu_walk_result) + 16, hipMemcpyDeviceToHost, f->wstream));
        // the alternative loci's walks on a stream of their own, beside the pair walks: 480 pair-walk wavefronts leave half
        // of the SIMDs without one, and one after the other the two launches were 4.1 + 1.9 ms of every call
        // (also wblk / the walk view they read: copied in on wstream -- astream waits for that copy below)
        if (n_alt > 0) {
            ICHK(f, hipStreamWaitEvent(f->astream, f->w0, 0));
            ICHK(f, launch_alt_walk(v, (const tredgpu_walk_task*)f->d_atask, (const tredgpu_walk_chunk*)(f->d_atask + n_alt * sizeof(tredgpu_walk_task)),
                                    (tredgpu_alt_result*)f->d_ares, f->d_ares + n_alt * sizeof(tredgpu_alt_result), (int)n_alt, f->astream));
            ICHK(f, hipEventRecord(f->aready, f->astream));
            ICHK(f, hipMemcpyAsync(f->h_ares, f->d_ares, n_alt * sizeof(tredgpu_alt_result) + nb, hipMemcpyDeviceToHost, f->astream));
        }
        ICHK(f, hipEventRecord(f->adone, f->astream));
        if (w->select && n_tasks > 0) {
            // the selection reads the pair walk's record lists (this stream) and the alternative loci's hits (the other one)
            if (n_alt > 0) ICHK(f, hipStreamWaitEvent(f->wstream, f->aready, 0));
            tredgpu_select_result* d_selres = (tredgpu_select_result*)(f->d_sel + n_tasks * sizeof(tredgpu_select_task));
            ICHK(f, launch_select(v, (const tredgpu_walk_task*)f->d_wtask, (const tredgpu_select_task*)f->d_sel,
                                  (const int64_t*)(f->d_wtask + n_tasks * sizeof(tredgpu_walk_task) + n_chunks * sizeof(tredgpu_walk_chunk)), f->d_wrecs,
                                  f->d_wfields, f->d_wchained, (const tredgpu_walk_result*)f->d_wres, (const tredgpu_walk_task*)f->d_atask,
                                  (const tredgpu_alt_result*)f->d_ares, (int)n_alt, f->d_sel_list, d_selres, (int)n_tasks, f->wstream));
            ICHK(f, hipEventRecord(f->w1, f->wstream));
            ICHK(f, hipMemcpyAsync(f->h_sel + n_tasks * sizeof(tredgpu_select_task), d_selres, n_tasks * sizeof(tredgpu_select_result), hipMemcpyDeviceToHost, f->wstream));
        }
        ICHK(f, hipEventRecord(f->wdone, f->wstream));
    }
    for (int s = 0; s < nstreams; ++s) {
        ICHK(f, hipEventRecord(f->t1[s], f->stream[s]));
        ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
    }
    tr.mark("enqueued");
    for (int s = 0; s < nstreams; ++s)
        if (wait_asleep(f, f->done[s])) return -10;
    tr.mark("decoded");
    int bad = 0;
    for (int32_t k = 0; k < n_blocks; ++k) { status[k] = f->h_status[k]; bad += status[k] != 0; }
    if (crc) for (int32_t k = 0; k < n_blocks; ++k) crc[k] = (uint32_t)f->h_status[f->cap_blocks + k];
    if (w) {
        if (wait_asleep(f, f->wdone)) return -10;
        tr.mark("pair_walk");
        if (wait_asleep(f, f->adone)) return -10;
        tr.mark("alt_walk");
        unsigned long long used[2];
        memcpy(used, f->h_wres + n_tasks * sizeof(tredgpu_walk_result), 16);
        memcpy(w->results, f->h_wres, n_tasks * sizeof(tredgpu_walk_result));
        if (w->select && n_tasks > 0) {
            memcpy(w->selected, f->h_sel + n_tasks * sizeof(tredgpu_select_task), n_tasks * sizeof(tredgpu_select_result));
            f->sel_tasks = (int)n_tasks;
        }
        if (n_alt > 0) {
            memcpy(w->alt_results, f->h_ares, n_alt * sizeof(tredgpu_alt_result));
            memcpy(w->need, f->h_ares + n_alt * sizeof(tredgpu_alt_result), (size_t)n_blocks);
        }
        // (a task that found its pool full took its room all the same: the counters can exceed the capacities)
        const size_t ng = (size_t)std::min<unsigned long long>(used[0], (unsigned long long)w->cap_global),
                     nt = (size_t)std::min<unsigned long long>(used[1], (unsigned long long)w->cap_target);
        for (int which = 0; which < 2; ++which) {
            int32_t** h = which ? &f->h_tpool : &f->h_gpool;
            size_t* cap = which ? &f->cap_htpool : &f->cap_hgpool;
            const size_t need = ((which ? nt : ng) + 16) * 4;
            if (need > *cap) {
                const size_t c = std::max(need + need / 8, *cap + *cap / 8);
                if (*h) (void)hipHostFree(*h);
                *h = nullptr; *cap = 0;
                ICHK(f, hipHostMalloc((void**)h, c, hipHostMallocDefault));
                *cap = c;
            }
        }
        if (ng) ICHK(f, hipMemcpyAsync(f->h_gpool, f->d_gpool, ng * 4, hipMemcpyDeviceToHost, f->wstream));
        if (nt) ICHK(f, hipMemcpyAsync(f->h_tpool, f->d_tpool, nt * 4, hipMemcpyDeviceToHost, f->wstream));
        ICHK(f, hipEventRecord(f->wdone, f->wstream));
        if (wait_asleep(f, f->wdone)) return -10;
        if (ng) memcpy(w->global_pool, f->h_gpool, ng * 4);
        if (nt) memcpy(w->target_pool, f->h_tpool, nt * 4);
        w->n_global = (int64_t)ng;
        w->n_target = (int64_t)nt;
        tr.mark("pools");
    }
    return bad;
}
}  // namespace

extern "C" {

int tredgpu_inflate_blocks_crc(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc) {
    return run_inflate(f, n_blocks, status, crc, true, nullptr);
}

int tredgpu_inflate_walk(tredgpu_inflater* f, int32_t n_blocks, int32_t* status, uint32_t* crc, tredgpu_walk_args* walk) {
    if (!walk) return f ? ifail(f, -2, "walk is NULL") : -2;
    return run_inflate(f, n_blocks, status, crc, false, walk);
}

// The blocks with need[k] != 0 of the last tredgpu_inflate_walk, copied to their places in the pinned output (runs of
// wanted blocks, and the unwanted ones between two runs when they are few, go in one copy).
int tredgpu_inflater_fetch(tredgpu_inflater* f, int32_t n_blocks, const uint8_t* need) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || (n_blocks > 0 && !need)) return ifail(f, -2, "bad arguments");
    if (n_blocks == 0) return 0;
    if (!f->h_out) return ifail(f, -2, "this inflater keeps no host copy of the output: tredgpu_inflater_fetch_dense");
    const int64_t* ooff = f->h_off + f->cap_blocks;
    ICHK(f, hipSetDevice(f->device));
    constexpr int64_t GAP = 128 * 1024;           // a copy costs the host ~5 us: less than these bytes cost the bus
    int copies = 0;
    int32_t k = 0;
    while (k < n_blocks) {
        if (!need[k]) { ++k; continue; }
        int32_t last = k;                          // the run [k, last]
        for (int32_t j = k + 1; j < n_blocks && ooff[j] - ooff[last + 1] <= GAP; ++j)
            if (need[j]) last = j;
        ICHK(f, hipMemcpyAsync(f->h_out + ooff[k], f->d_out + ooff[k], (size_t)(ooff[last + 1] - ooff[k]), hipMemcpyDeviceToHost, f->stream[copies & 1]));
        ++copies;
        k = last + 1;
    }
    for (int s = 0; s < 2; ++s) {
        ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
        if (wait_asleep(f, f->done[s])) return -10;
    }
    return copies;
}

int tredgpu_inflate_blocks(tredgpu_inflater* f, int32_t n_blocks, int32_t* status) { return tredgpu_inflate_blocks_crc(f, n_blocks, status, nullptr); }

int tredgpu_inflater_timing(tredgpu_inflater* f, double* total_ms, double* kernel_ms) {
    if (!f || !total_ms || !kernel_ms) return -2;
    *total_ms = *kernel_ms = 0.0;
    if (f->last_slices == 0) return 0;
    ICHK(f, hipSetDevice(f->device));
    float ms = 0.f;
    for (int s = 0; s < f->last_streams; ++s) {
        ICHK(f, hipEventElapsedTime(&ms, f->t0[0], f->t1[s]));
        *total_ms = std::max(*total_ms, (double)ms);
    }
    for (int k = 0; k < f->last_slices; ++k) {
        if (hipEventElapsedTime(&ms, f->k0[k], f->k1[k]) == hipSuccess) *kernel_ms += ms;
    }
    return 0;
}

int tredgpu_inflater_host_out(tredgpu_inflater* f, int enabled) {
    if (!f) return -2;
    if ((enabled != 0) != f->host_out) {
        ICHK(f, hipSetDevice(f->device));
        for (hipStream_t st : f->stream) ICHK(f, hipStreamSynchronize(st));
        release(f);                                    // (the next reserve allocates what the new mode needs)
        f->host_out = enabled != 0;
    }
    return 0;
}

int tredgpu_inflater_fetch_dense(tredgpu_inflater* f, int32_t n_blocks, const uint8_t* need, uint8_t** host, int64_t* dense_off) {
    if (!f) return -2;
    if (n_blocks < 0 || (size_t)n_blocks + 1 > f->cap_blocks || !host || !dense_off || (n_blocks > 0 && !need)) return ifail(f, -2, "bad arguments");
    CallTrace tr("fetch_dense");
    const int64_t* ooff = f->h_off + f->cap_blocks;
    constexpr int64_t GAP = 128 * 1024;           // a copy costs the host ~5 us: less than these bytes cost the bus
    // the runs: a wanted block, and on to the next wanted one while the blocks in between are fewer bytes than GAP
    struct Run { int32_t first, last; };
    std::vector<Run> runs;
    int64_t total = 0;
    dense_off[0] = 0;
    int32_t k = 0;
    while (k < n_blocks) {
        if (!need[k]) { dense_off[k + 1] = total; ++k; continue; }
        int32_t last = k;
        for (int32_t j = k + 1; j < n_blocks && ooff[j] - ooff[last + 1] <= GAP; ++j)
            if (need[j]) last = j;
        for (int32_t j = k; j <= last; ++j) { total += ooff[j + 1] - ooff[j]; dense_off[j + 1] = total; }
        runs.push_back(Run{k, last});
        k = last + 1;
    }
    ICHK(f, hipSetDevice(f->device));
    if ((size_t)total + 64 > f->cap_dense) {
        const size_t c = std::max((size_t)total + 64, f->cap_dense + f->cap_dense / 8);
        if (f->h_dense) (void)hipHostFree(f->h_dense);
        f->h_dense = nullptr; f->cap_dense = 0;
        ICHK(f, hipHostMalloc((void**)&f->h_dense, c, hipHostMallocDefault));
        f->cap_dense = c;
    }
    tr.mark("room");
    *host = f->h_dense;
    int copies = 0;
    static const bool by_dma = getenv("TREDGPU_FETCH_DMA") != nullptr;     // (A/B: the copy engines, one copy per run)
    if (by_dma) {
        for (const Run& r : runs) {
            ICHK(f, hipMemcpyAsync(f->h_dense + dense_off[r.first], f->d_out + ooff[r.first], (size_t)(ooff[r.last + 1] - ooff[r.first]),
                                   hipMemcpyDeviceToHost, f->stream[copies & 1]));
            ++copies;
        }
        tr.mark("enqueued");
        for (int s = 0; s < 2; ++s) {
            ICHK(f, hipEventRecord(f->done[s], f->stream[s]));
            if (wait_asleep(f, f->done[s])) return -10;
        }
    } else if (!runs.empty()) {
        constexpr int64_t PIECE = 32 * 1024;       // bytes per workgroup: ~3 300 workgroups for a 16-sample call
        size_t n_pieces = 0;
        for (const Run& r : runs) n_pieces += (size_t)((ooff[r.last + 1] - ooff[r.first] + PIECE - 1) / PIECE);
        if (n_pieces > f->cap_pieces) {
            const size_t c = std::max(n_pieces, f->cap_pieces + f->cap_pieces / 2);
            if (f->h_pieces) (void)hipHostFree(f->h_pieces);
            if (f->d_pieces) (void)hipFree(f->d_pieces);
            f->h_pieces = f->d_pieces = nullptr; f->cap_pieces = 0;
            ICHK(f, hipHostMalloc((void**)&f->h_pieces, c * sizeof(FetchPiece), hipHostMallocDefault));
            ICHK(f, hipMalloc((void**)&f->d_pieces, c * sizeof(FetchPiece)));
            f->cap_pieces = c;
        }
        FetchPiece* P = (FetchPiece*)f->h_pieces;
        size_t p = 0;
        for (const Run& r : runs) {
            const int64_t bytes = ooff[r.last + 1] - ooff[r.first];
            for (int64_t o = 0; o < bytes; o += PIECE) P[p++] = FetchPiece{ooff[r.first] + o, dense_off[r.first] + o, (int32_t)std::min(PIECE, bytes - o), 0};
        }
        ICHK(f, hipMemcpyAsync(f->d_pieces, f->h_pieces, n_pieces * sizeof(FetchPiece), hipMemcpyHostToDevice, f->stream[0]));
        ICHK(f, launch_fetch_gather(f->d_out, f->h_dense, (const FetchPiece*)f->d_pieces, n_pieces, f->stream[0]));
        copies = (int)n_pieces;
        tr.mark("enqueued");
        ICHK(f, hipEventRecord(f->done[0], f->stream[0]));
        if (wait_asleep(f, f->done[0])) return -10;
    }
    tr.mark("copied");
    if (tr.on) { char b[64]; snprintf(b, sizeof b, " copies=%d MB=%.1f", copies, total / 1e6); tr.line += b; }
    return copies;
}

int64_t tredgpu_inflater_pinned_bytes(const tredgpu_inflater* f) {
    if (!f) return -2;
    size_t n = f->cap_comp + (f->h_out ? f->cap_out : 0) + f->cap_blocks * (2 * sizeof(int64_t) + 2 * sizeof(int32_t)) + f->cap_dense +
               f->cap_pieces * sizeof(FetchPiece) + f->cap_wblk + f->cap_wtask + f->cap_wres + f->cap_hgpool + f->cap_htpool + f->cap_atask + f->cap_ares +
               f->cap_sel;
    return (int64_t)n;
}

int64_t tredgpu_inflater_walk_serial_regions(tredgpu_inflater* f) {
    if (!f) return -2;
    if (f->last_walk_tasks == 0 || !f->d_wchained) return 0;
    ICHK(f, hipSetDevice(f->device));
    std::vector<WalkChained> c(f->last_walk_tasks);
    ICHK(f, hipMemcpy(c.data(), f->d_wchained, c.size() * sizeof(WalkChained), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (const WalkChained& w : c) n += w.mode == 0;
    if (getenv("TREDGPU_TRACE") != nullptr)
        for (size_t t = 0; t < c.size(); ++t)
            if (c[t].mode == 0) fprintf(stderr, "tredgpu: region %zu chained serially (reason %d), %d records, status %d\n", t, c[t].pad, c[t].n, c[t].status);
    return n;
}

int tredgpu_inflater_walk_ms(tredgpu_inflater* f, double* walk_ms) {
    if (!f || !walk_ms) return -2;
    *walk_ms = 0.0;
    if (!f->walk_timed) return 0;
    ICHK(f, hipSetDevice(f->device));
    float ms = 0.f;
    ICHK(f, hipEventElapsedTime(&ms, f->w0, f->w1));
    *walk_ms = ms;
    return 0;
}

}  // extern "C"

// what tredgpu_genotype_selected (capi.hip) reads of an inflater: the decoder's output, the selected records' places and the
// host copy of the select results of its last walk
int tredgpu_front::inflater_selected(tredgpu_inflater* f, SelectedView* view) {
    if (!f || !view) return -2;
    if (f->sel_tasks <= 0 || !f->d_sel_list || !f->d_out) return ifail(f, -2, "the inflater's last call carried no read selection");
    view->device = f->device;
    view->out = f->d_out;
    view->sel_list = f->d_sel_list;
    view->n_tasks = f->sel_tasks;
    view->results = (const tredgpu_select_result*)(f->h_sel + (size_t)f->sel_tasks * sizeof(tredgpu_select_task));
    return 0;
}
