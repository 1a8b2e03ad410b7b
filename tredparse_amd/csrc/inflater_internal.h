// inflater_internal.h -- what the three translation units of the front end share (inflate_decode.hip: the DEFLATE decoder;
// walk.hip: the record walks and the fetch kernel over its output; inflater_api.hip: the C ABI of include/tredgpu.h section 4):
// the walk's views and tuples, and one launcher per kernel -- kernels stay local to their file, as in sw_ladder.hip / grid.hip.
#ifndef TREDGPU_INFLATER_INTERNAL_H
#define TREDGPU_INFLATER_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "../../include/tredgpu.h"

namespace tredgpu_front {

constexpr int LANES = 64;

__device__ __forceinline__ int wave_incl_scan(int v) {   // all 64 lanes active
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr 1, 2, 4, 8: prefix inside a 16-lane row
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast 15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast 31 into rows 2 and 3
    return v;
}

struct WalkView {
    const uint8_t* out; const int64_t* ooff;          // the decoder's output and its block offsets
    const int32_t* bstatus; const uint32_t* bcrc;     // what the decoder said about each block
    const uint32_t* xcrc; const int64_t* bcoff; const int32_t* bclen;   // from the file: trailer CRC, compressed offset / length
    int64_t out_end;                                  // bytes of `out` that may be read
};
struct WalkPair { int64_t name_at, name2_at; int32_t a_pos, a_lead, b_end, b_trail; uint16_t name_len; uint8_t a_rev, b_rev, complete, pad[3]; };
static_assert(sizeof(WalkPair) == 40, "WalkPair layout");
constexpr int WALK_PAIR_CAP = 8192;               // names per region at most (a +-10 kb window at 30x holds ~2 100) ...
constexpr int WALK_PAIR_CAP_SMALL = 4096;         // ... and what a launch whose regions are all short is given: 38 instead of
                                                  // 70 KB of LDS per wavefront, so that other kernels' workgroups -- the
                                                  // decoder's, the genotyping kernels' of the other driver processes -- still
                                                  // find LDS on the CUs a walk occupies (two walks of 70 KB nearly fill a CU's 160 KB)
constexpr int WALK_WINDOW = 6144;                 // bytes of the block stream in LDS
enum { WALK_OK = 0, WALK_NOT_PLANNED = 1, WALK_BAD_BLOCK = 2, WALK_BAD_RECORD = 3, WALK_TABLE_FULL = 4, WALK_NO_END = 5, WALK_POOL_FULL = 6,
       WALK_TAG_CLASH = 7 };

struct WalkRec { int64_t a0; uint64_t at, after; };             // where the record's length word lies in `out`; the virtual offsets of the record and of what follows it
struct WalkFields { uint32_t h; int32_t rtid, rpos, rend, lead, trail; uint16_t flag, nlen; uint32_t bad; };
static_assert(sizeof(WalkRec) == 24 && sizeof(WalkFields) == 32, "record tuples");
struct WalkChained { int32_t status, n, mode, klo, khi, pad; };   // per region: how the chain ended, records listed; mode 1: listed by
                                                                  // walk_chain_par_kernel (the region's blocks lie in [klo, khi)), 0: by
                                                                  // walk_chain_kernel (pad: why the lanes handed the region back)

struct FetchPiece { int64_t src, dst; int32_t len, pad; };

constexpr size_t walk_lds_bytes(int cap) { return (size_t)cap * 16; }    // (pair_walk_kernel: the table alone, 2 * cap slots of 64 bits)

constexpr int PW_WAVES = 8, PW_THREADS = PW_WAVES * LANES;      // pair_walk_kernel: a workgroup of 8 wavefronts per region

// TREDGPU_CTX_CUS=N (tuning / A-B): the device's compute units are split between the front end's streams
// (decode, walks: the first CUs - N bits of the CU mask) and the genotyping context's stream (the last N) -- a genotyping
// call's dozen short launches then never queue behind a device full of decoder wavefronts, at the price of the decoder's
// share.  Unset or 0: every stream may use every CU (the default: the front end is decode-bound).
// which: 0 front end, 1 context.  Returns hipSuccess with *st created (plainly, with `flags` and `priority`, when no split is set).
inline hipError_t create_partitioned_stream(hipStream_t* st, unsigned flags, int priority, int which, int device) {
    const char* want = getenv("TREDGPU_CTX_CUS");
    const int n_ctx = want ? atoi(want) : 0;
    hipDeviceProp_t prop;
    if (n_ctx > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && n_ctx < prop.multiProcessorCount) {
        const int cus = prop.multiProcessorCount;
        uint32_t mask[16] = {};
        for (int k = 0; k < cus && k < 512; ++k) {
            const bool ctx_cu = k >= cus - n_ctx;
            if (ctx_cu == (which == 1)) mask[k >> 5] |= 1u << (k & 31);
        }
        return hipExtStreamCreateWithCUMask(st, (uint32_t)((cus + 31) / 32), mask);
    }
    return hipStreamCreateWithPriority(st, flags, priority);
}

// ---- launchers (each enqueues on `st` and returns hipGetLastError()) ----------------------------------------------------
// inflate_decode.hip: blocks [first_block, first_block + n_blocks) of the call, one wavefront each
hipError_t launch_inflate(const uint32_t* comp, const int64_t* comp_off, uint8_t* out, const int64_t* out_off, int first_block, int n_blocks,
                          int32_t* status, uint32_t* crc_out, hipStream_t st);
// walk.hip
hipError_t launch_walk_chain_par(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, const int64_t* rec_base,
                                 WalkRec* recs, WalkChained* chained, int n_tasks, hipStream_t st);
hipError_t launch_walk_chain(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, const int64_t* rec_base,
                             WalkRec* recs, WalkChained* chained, int n_tasks, hipStream_t st);
hipError_t launch_walk_parse(const WalkView& v, int n_tasks, const int64_t* rec_base, const WalkRec* recs, const WalkChained* chained,
                             WalkFields* fields, size_t total_recs, hipStream_t st);
hipError_t allow_pair_walk_lds();        // once per device: the large table's dynamic LDS
hipError_t launch_pair_walk(const WalkView& v, const tredgpu_walk_task* tasks, const int64_t* rec_base, const WalkRec* recs, const WalkFields* fields,
                            const WalkChained* chained, tredgpu_walk_result* results, WalkPair* pairs, int32_t* gpool, int64_t cap_g, int32_t* tpool,
                            int64_t cap_t, unsigned long long* counters, int table_cap, int n_tasks, hipStream_t st);
hipError_t launch_alt_walk(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_walk_chunk* chunks, tredgpu_alt_result* results,
                           uint8_t* need, int n_tasks, hipStream_t st);
hipError_t launch_fetch_gather(const uint8_t* out, uint8_t* host, const FetchPiece* pieces, size_t n_pieces, hipStream_t st);
// the read selection over the record lists of the pair walk (tredgpu.h section 5); sel_list: TREDGPU_SELECT_CAP places per task
hipError_t launch_select(const WalkView& v, const tredgpu_walk_task* tasks, const tredgpu_select_task* sel, const int64_t* rec_base,
                         const WalkRec* recs, const WalkFields* fields, const WalkChained* chained, const tredgpu_walk_result* results,
                         const tredgpu_walk_task* alt_tasks, const tredgpu_alt_result* alt_results, int n_alt_tasks, int64_t* sel_list,
                         tredgpu_select_result* out, int n_tasks, hipStream_t st);
// units [u0, u0 + n_units) of a batch (unit_* arrays indexed by batch unit) from one inflater's selection into the batch's arrays
hipError_t launch_pack_selected(const uint8_t* out, const int64_t* sel_list, const int32_t* unit_task, const int32_t* unit_read_off,
                                const int64_t* unit_word_off, const int64_t* unit_seq4_off, const int64_t* unit_name_off, int u0, int n_units,
                                uint32_t* packed, int64_t* read_off, int32_t* read_len, uint8_t* seq4, int64_t* seq4_off, uint8_t* names,
                                int64_t* name_off, hipStream_t st);

// inflater_api.hip: what tredgpu_genotype_selected (capi.hip) reads of an inflater whose last walk carried a selection
struct SelectedView { int device; const uint8_t* out; const int64_t* sel_list; int n_tasks; const tredgpu_select_result* results; };
int inflater_selected(tredgpu_inflater* inf, SelectedView* view);    // 0, or -2 when the last call had no selection

}  // namespace tredgpu_front
#endif
