// Internal declarations shared by the HIP translation units of libtredgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tredgpu.h"

namespace tredgpu {

// One template ladder on the device.  Sequences are base codes 0..4 (N = 4) in `seq`:
//   strand 0: trunk = prefix + repeat*max_units, branch = suffix              (bam_parser.py:92)
//   strand 1: trunk = rc(suffix) + rc(repeat)*max_units, branch = rc(prefix)  (bam_parser.py:93)
// max_units == 0: a plain reference (trunk = the sequence, no branch, strand 0 only).
struct LadderDesc {
    int32_t alen[2];       // trunk head length per strand (|prefix| / |suffix|)
    int32_t blen[2];       // branch length per strand     (|suffix| / |prefix|)
    int32_t trunk_off[2];  // WORD offsets into the letter pool (8 letters per 32-bit word)
    int32_t branch_off[2];
    int32_t period;
    int32_t max_units;
    int32_t n_strands;
    int32_t kmer_ok;       // 1: kmer_off[] bitmaps usable (max_units > 0)
    int32_t kmer_off[2];   // WORD offsets of the 4096-bit 6-mer presence bitmaps (128 words) per strand
};

// A quad = up to four reads of one ladder (any units) that share a wavefront (16 lanes each).
struct Quad {
    int32_t ladder;
    int32_t first;    // index into the permutation array: reads perm[first .. first+count)
    int32_t count;
    int32_t strands;  // bit s set: strand s has to be swept (all reads of a quad share the class)
};

struct SwArgs {
    const uint32_t* packed;
    const int64_t* read_off;
    const int32_t* read_len;
    const int32_t* unit_read_off;
    const int32_t* unit_ladder;
    const LadderDesc* ladders;
    const uint32_t* seqw;  // ladder letters, 8 per word (4 bits each)
    const Quad* quads;
    const int32_t* perm;     // class-sorted read order (see build_quads)
    const int32_t* n_quads;  // device counter written by build_quads
    uint8_t* out_tag;
    int16_t* out_h;
    int16_t* out_score;
    int16_t* out_dump;
    int32_t dump_templates;
    int32_t n_units;
    int32_t n_ladders;       // registered ladders (unit_ladder values outside [0, n_ladders) are clamped on the device)
    int32_t max_rows;        // 16 * rows-per-lane of the instantiation that will run
    tredgpu_sw_params p;
    unsigned long long* stats;  // [SW_STAT_SLOTS][8], slot = workgroup & (SW_STAT_SLOTS - 1): work counters: trunk cols, continuation-pass cols, templates combined/dropped/emitted-from-trunk, waves
};

// per-wave counter updates are spread over this many 64-byte lines: half a million waves adding to one line
// serialise in L2 (measured: 6 atomics per wave on one line cost the SW kernel 15 % of its time)
constexpr int SW_STAT_SLOTS = 1024;

// sw_ladder.hip
hipError_t launch_build_quads(const SwArgs& a, uint8_t* read_class, int32_t* perm, Quad* quads, int32_t* n_quads,
                              int32_t* unit_cnt, int32_t* bins, int n_ladders, int64_t max_quads, hipStream_t s);
size_t sw_bin_bytes(int n_ladders);
size_t sw_unit_cnt_bytes(int n_units);
int64_t sw_max_quads(int64_t n_reads, int n_ladders);   // one partial quad per (ladder, class, level) bin
// generic: some ladder's branch could reach the score filter on its own (the production kernel skips that sweep)
hipError_t launch_sw_ladder(const SwArgs& a, int rows_per_lane, bool generic, int64_t max_quads, hipStream_t s);
hipError_t launch_tally(const uint8_t* tag, const int16_t* h, int64_t n_reads,
                        const int32_t* unit_read_off, int32_t n_units, const int32_t* read_pair_id,
                        int32_t hist_stride, int32_t* full_cnt, int32_t* pref_cnt,
                        int32_t* rept_cnt, uint8_t* scratch_drop, hipStream_t s);

// grid.hip
struct ModelConst {
    double step[6][37];
    double w[5];
    double gc, score;
    double small;         // SMALL_VALUE = exp(-10), models.py:34 (host libm value)
    double really_small;  // REALLY_SMALL_VALUE = exp(-100), models.py:35
    double logsmall;      // log(SMALL_VALUE)
    // gammaln(n + 1) for the repeat-only read counts (models.py:215, scipy poisson.pmf): host lgamma for
    // n < GRID_LFACT, Stirling's series beyond (exact to the last bit there)
    double lfact[4096];
};
constexpr int GRID_LFACT = 4096;

struct GridArgs {
    const tredgpu_unit_params* units;
    int32_t n_units;
    int32_t hist_stride;
    const int32_t* full_cnt;
    const int32_t* pref_cnt;
    const int32_t* rept_cnt;
    const int32_t* global_lens;
    const int32_t* target_lens;
    tredgpu_call* calls;
    const int64_t* grid_off;
    double* grid_dump;
    double* marg;
    int32_t marg_stride;
    const ModelConst* model;
    const int64_t* joint_off;   // sparse joint distribution (optional): capacities, in triples
    double* joint;              // {h1, h2, exp(ml - max)} per kept pair
    int32_t* joint_n;           // qualifying pairs per unit
    double* joint_total;        // sum over the unit's distinct pairs
    double* kde_pdf;      // tredgpu_pe_kde only: [n_units][1000] output
    int32_t* kde_status;  // tredgpu_pe_kde only: [n_units]
    double* unit_pdf;      // grid passes: [n_units][1000] KDEs of the units whose paired-end term is used (grid_kde_kernel)
    int32_t* unit_kde_rc;  // grid passes: [n_units] outcome of the unit's KDE (0, -2, -6)
    int32_t max_target;    // grid passes: largest n_target of the batch (bounds a unit's slot, grid_subpools)
};
constexpr int GRID_MAX_ROWS = 1024;  // |h1range| / |h2range| the grid kernels accept (status -5 beyond)
constexpr int GRID_MAX_COLS = 1024;
constexpr size_t GRID_POOL_BYTES = (size_t)12 << 30;  // scratch pool the units' tables are carved from
constexpr int GRID_UNIT_DEFERRED = 100;               // calls[].status of a unit waiting for the next pass

hipError_t launch_pe_kde(const GridArgs& a, hipStream_t s);
// max over units of maxinsert -> out[0], of n_target -> out[1] (device ints, zeroed by the launch)
hipError_t launch_unit_max(const tredgpu_unit_params* units, int n_units, int* out, hipStream_t s);
size_t grid_desc_bytes();
size_t grid_counter_bytes();
int grid_deferred_offset();                          // byte offset of the deferred-unit count in the counter block
size_t grid_slot_doubles_max(int rows_cap, int cols_cap, int nt_max);
size_t grid_items_cap(int rows_cap, int cols_cap);   // work items one unit can make at most
size_t grid_item_bytes();                            // bytes per work item in the items buffer
size_t grid_item_slots(int n_units, int rows_cap, int cols_cap);   // entries of the work-item list (16 regions, one per ticket queue)
int grid_subpools(size_t pool_doubles, int rows_cap, int cols_cap, int nt_max);   // sub-pools the scratch pool is used as
size_t grid_units_per_subpool(int n_units, int n_sub);
// One pass over all units (prepare -> pairs -> reduce); see grid.hip
hipError_t launch_grid_pass(const GridArgs& a, int pass, void* descs, double* pool, size_t pool_doubles, int rows_cap,
                            int cols_cap, void* items, size_t item_cap, void* counters, hipStream_t s, int phases = 15);

}  // namespace tredgpu
