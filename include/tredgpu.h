/*
 * tredgpu.h -- C ABI of libtredgpu.so, the MI355X (gfx950) STR-genotyping hot path.
 *
 * This is the drop-in boundary for the two compute steps of humanlongevity/tredparse
 * (reference paths relative to /root/reference):
 *
 *   (1) per-read Smith-Waterman against the template ladder + read tagging
 *       replaces   src/ssw.h:72-182 (ssw_init / ssw_align / init_destroy / align_destroy),
 *                  bound one alignment at a time by src/ssw_wrap.py:69-83,177-227, and the loop
 *                  around it in tredparse/bam_parser.py:84-182 (_buildDB, get_hangs, _parseReadSW)
 *                  and :256-268 (tally_counts, rept).
 *   (2) the (h1,h2) allele-pair likelihood grid
 *       replaces   tredparse/models.py:149-302 (pdf_spanning .. evaluate), :319-368 (calc_CI,
 *                  calc_PP) and :426-473 (PEMaxLikModel incl. the gaussian_kde call).
 *
 * Conventions
 *   - plain C, no C++/torch types; every buffer is caller-owned; the library allocates only inside
 *     the opaque context.  No function calls exit() or throws (contrast ssw.c:584-587).
 *   - every function returns 0 on success, <0 on error; tredgpu_last_error() gives the text.
 *   - mem: TREDGPU_MEM_HOST -> array arguments are host pointers (the call copies in, runs,
 *     copies out and synchronises); TREDGPU_MEM_DEVICE -> they are device pointers valid on the
 *     context's GPU (e.g. torch tensors' data_ptr()); the call only enqueues work on the context's
 *     stream and returns; use tredgpu_sync() or stream-ordered consumers.
 *   - a context is bound to one GPU and one HIP stream; calls on one context are serialised on that
 *     stream; different contexts are independent (one process per GPU).
 *   - "unit" = one sample x locus group (one runBam call, tredparse/tred.py:153-169).
 *   - "ladder" = the template set of one locus at one READLEN: prefix + repeat*u + suffix and its
 *     reverse complement for u = 1..max_units (bam_parser.py:84-100).
 */
#ifndef TREDGPU_H
#define TREDGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TREDGPU_MEM_HOST 0
#define TREDGPU_MEM_DEVICE 1

/* read tags, bam_parser.py:157-168 (POST shares PREF's histogram, bam_parser.py:77) */
#define TREDGPU_TAG_NONE 0
#define TREDGPU_TAG_FULL 1
#define TREDGPU_TAG_PREF 2
#define TREDGPU_TAG_POST 3
#define TREDGPU_TAG_REPT 4
#define TREDGPU_TAG_HANG 5
#define TREDGPU_TAG_INVALID 255 /* read longer than the instantiated kernel handles */

#define TREDGPU_MAX_READ_LEN 480   /* longest read the kernels are instantiated for (512 rows: the packed values' nine row bits) */
#define TREDGPU_ASSUMED_READ_LEN 320 /* the bound a call in DEVICE memory gets when it names none (max_read_len 0) */
#define TREDGPU_MAX_TEMPLATE_LEN 511
#define TREDGPU_SPAN 1000          /* SPAN, bam_parser.py:29 / models.py pdf length */

typedef struct tredgpu_ctx tredgpu_ctx;

/* scoring + tagging constants: bam_parser.py:95-98 (1/5/7/2), :30 (FLANKMATCH 9), :154-155 (clip).
 * Accepted range: match 1..8, mismatch 0..16, 1 <= gap_extend <= gap_open <= 16, and -- because a DP value is
 * (score + (row + col) * gap_extend) << 18 | start cell in one int32 --
 *     (rows + 511) * gap_extend + max_read_len * match < 8192,   rows = 64 / 112 / 160 / 256 / 320 / 512 for
 * max_read_len (0: 320 for DEVICE memory, the longest read of the call for HOST memory).  Anything else is refused with status -2; 1/5/7/2 needs 1 790 (2 526 at 480 bp). */
typedef struct tredgpu_sw_params {
    int32_t match;      /* +match on the diagonal                (ssw_wrap.py:154-167) */
    int32_t mismatch;   /* -mismatch off the diagonal, N scores 0                       */
    int32_t gap_open;   /* first gap base costs gap_open          (ssw.c:225-232)        */
    int32_t gap_extend; /* every further gap base costs gap_extend                      */
    int32_t flank;      /* FLANKMATCH                                                   */
    int32_t clip;       /* --useclippedreads: REPT cut-off from the read's own length   */
    int32_t max_read_len; /* upper bound of read_len[] in the call (selects the kernel
                             instantiation: 64/112/160/256/320/512 rows); 0 = scan read_len (HOST
                             memory) or assume TREDGPU_ASSUMED_READ_LEN (DEVICE memory).  A read
                             longer than the bound gets out_tag TREDGPU_TAG_INVALID.      */
    int32_t reserved;
} tredgpu_sw_params;

/* ---- life cycle ------------------------------------------------------------------------- */
int tredgpu_create(int device_id, tredgpu_ctx** out);
void tredgpu_destroy(tredgpu_ctx* ctx);
const char* tredgpu_last_error(const tredgpu_ctx* ctx);  /* ctx may be NULL: last create error */
int tredgpu_sync(tredgpu_ctx* ctx);
/* the HIP stream (hipStream_t) all work of this context is enqueued on */
void* tredgpu_get_stream(tredgpu_ctx* ctx);
/* "tredgpu <ver> (gfx950) src <16 hex digits>": the hash of the kernel sources this library was built from (csrc/Makefile);
 * the PMC summaries under profiles/ carry the same hash and bench.py cites their counters only when the two agree */
const char* tredgpu_version(void);

/* ---- static tables ------------------------------------------------------------------------ */
/*
 * Register the template ladders (replaces BamParser._buildDB, bam_parser.py:84-100, which the
 * reference rebuilds for every sample x locus).  Ladder i = (prefix[i], repeat[i], suffix[i],
 * max_units[i]); sequences are ASCII, N allowed (motifs such as GCN), case-insensitive.
 * max_units[i] = ceil(READLEN / len(repeat)) (bam_parser.py:73); 0 registers a plain reference
 * sequence (prefix only) for tredgpu_sw_dump-style single alignments (Aligner.align).
 * Replaces any previously registered set.
 */
int tredgpu_set_ladders(tredgpu_ctx* ctx, int32_t n_ladders, const char* const* prefix,
                        const char* const* repeat, const char* const* suffix,
                        const int32_t* max_units);

/*
 * Stutter / step-size model constants (replaces StepModel / NoiseModel, models.py:42-84).
 *   step_pdf: 6 x 37 doubles, row p-1 = "Period<p>Model" (index 18 = offset 0);
 *             periods 7..17 reuse row 6 (models.py:59-60); period >= 18 is an error as in the
 *             reference (KeyError).
 *   stutter_w: 5 doubles: bias, w_period, w_units, w_gc, w_score (models.py:79-84).
 */
int tredgpu_set_model(tredgpu_ctx* ctx, const double* step_pdf, const double* stutter_w,
                      double gc, double score);

/* ---- host-side packing helper ------------------------------------------------------------- */
/*
 * 2-bit + N-mask packing of reads (replaces Aligner._DNA_to_int_mat, ssw_wrap.py:229-244).
 * seqs: concatenated ASCII reads, seq_off[n_reads+1] byte offsets.  For read r of length L the
 * packed record is ceil(L/16) words of 2-bit codes (A0 C1 G2 T3, base i at bits 2*(i%16) of word
 * i/16) followed by ceil(L/32) words of N flags (bit i%32 of word i/32; any non-ACGT letter).
 * word_off_out[n_reads+1] receives the record offsets (in 32-bit words), len_out[n_reads] the
 * lengths.  Returns the total number of words, or <0; call with packed_out == NULL to size.
 */
int64_t tredgpu_pack_reads(const char* seqs, const int64_t* seq_off, int64_t n_reads,
                           uint32_t* packed_out, int64_t* word_off_out, int32_t* len_out);

/* ---- (1) template Smith-Waterman + tagging ------------------------------------------------- */
/*
 * For every read: align against all 2*max_units templates of its unit's ladder, filter
 * (ssw_wrap.py:214-220 with min_len/min_score of bam_parser.py:133-134), tag (bam_parser.py:139-168)
 * and keep max(key=(score,-units)) (bam_parser.py:174).
 *
 *   packed, read_off[n_reads+1], read_len[n_reads]   as produced by tredgpu_pack_reads
 *   unit_read_off[n_units+1]   reads of unit g are [unit_read_off[g], unit_read_off[g+1])
 *   unit_ladder[n_units]       ladder index of each unit
 *   out_tag[n_reads] (TREDGPU_TAG_*), out_h[n_reads] (repeat units), out_score[n_reads]
 *   out_dump: optional (NULL to skip), int16 [n_reads][dump_templates][6] =
 *             {score, ref_begin, ref_end, read_begin, read_end, tag-before-argmax} per template in
 *             the reference's db order (u=1 fwd, u=1 rc, u=2 fwd, ...), i.e. field-for-field the
 *             s_align members ssw_wrap.py:302-310 copies; -1 rows for templates a ladder lacks.
 */
int tredgpu_sw_classify(tredgpu_ctx* ctx, int mem, const uint32_t* packed, const int64_t* read_off,
                        const int32_t* read_len, int64_t n_reads, const int32_t* unit_read_off,
                        const int32_t* unit_ladder, int32_t n_units,
                        const tredgpu_sw_params* params, uint8_t* out_tag, int16_t* out_h,
                        int16_t* out_score, int16_t* out_dump, int32_t dump_templates);

/*
 * Histograms of one batch (replaces tally_counts + rept, bam_parser.py:256-268):
 *   full_cnt[n_units][hist_stride], pref_cnt[...] (PREF and POST together, bam_parser.py:77),
 *   rept_cnt[...]; bin = repeat units h (< hist_stride).  All int32.
 * If read_pair_id != NULL (int32 per read, equal for the two mates of a pair, <0 = none) pairs whose
 * reads are both REPT are dropped first (remove_pairs_of_rept, bam_parser.py:270-287; --norepeatpairs).
 */
int tredgpu_tally(tredgpu_ctx* ctx, int mem, const uint8_t* tag, const int16_t* h, int64_t n_reads,
                  const int32_t* unit_read_off, int32_t n_units, const int32_t* read_pair_id,
                  int32_t hist_stride, int32_t* full_cnt, int32_t* pref_cnt, int32_t* rept_cnt);

/* ---- (2) likelihood grid -------------------------------------------------------------------- */
/* per-unit inputs of IntegratedCaller (models.py:106-147) that are not histograms */
typedef struct tredgpu_unit_params {
    int32_t period;        /* len(repeat)                                  models.py:113   */
    int32_t readlen;       /* READLEN                                      models.py:112   */
    int32_t ploidy;        /* 1 or 2                                       models.py:122   */
    int32_t maxinsert;     /* --maxinsert (300)                            models.py:124   */
    int32_t fullsearch;    /* --fullsearch                                 models.py:125   */
    int32_t ref_len;       /* repeat_end - repeat_start + 1                bam_parser.py:68 */
    int32_t minpe;         /* end - start + 2*FLANKMATCH + 2               bam_parser.py:361 */
    int32_t cutoff_risk;   /* tred.cutoff_risk                             models.py:342-368 */
    int32_t is_expansion;  /* mutation_nature == 'increase'                meta.py:125     */
    int32_t is_recessive;  /* inheritance[-1] == 'R'                       meta.py:124     */
    int32_t pe_off;        /* this unit's slice of global_lens: [pe_off, pe_off+n_global) */
    int32_t n_global;      /* len(pe.global_lens)                          bam_parser.py:342-359 */
    int32_t tl_off;        /* this unit's slice of target_lens                             */
    int32_t n_target;      /* len(pe.target_lens)                                          */
    double half_depth;     /* depth / 2                                    models.py:123   */
} tredgpu_unit_params;

/* per-unit outputs of IntegratedCaller.evaluate (models.py:223-302) */
typedef struct tredgpu_call {
    int32_t status;     /* 0 ok; 1 no evidence (alleles -1,-1; models.py:244-245,406-408);
                           <0 error that the reference would raise for this unit (the locus is
                           dropped by tred.py:245-249): -2 singular KDE, -3 observation >= SPAN */
    int32_t n_pairs;    /* size of the evaluated grid                                    */
    int32_t h1, h2;     /* arg-max pair in bp (key (ml, -h1), first in enumeration; models.py:299) */
    int32_t ci[4];      /* h1_lo, h1_hi, h2_lo, h2_hi in repeat units (models.py:287-290) */
    int32_t run_pe;     /* models.py:234-236                                             */
    int32_t pad;
    double lik;         /* max log-likelihood                                            */
    double pp;          /* calc_PP, models.py:342-368                                    */
} tredgpu_call;

/*
 * Evaluate the grid of every unit.
 *   full_cnt / pref_cnt / rept_cnt: as produced by tredgpu_tally (rept = sum of a unit's REPT bins)
 *   global_lens / target_lens: int32 pools indexed through pe_off/tl_off (PEextractor output)
 *   calls[n_units]
 *   dump (optional): grid_off[n_units+1] (int64, caller-computed capacity offsets) and
 *   grid_dump = doubles [total][6] = {h1, h2, ml1, ml2, ml3, ml4} in enumeration order
 *   (models.py:260-273); pass NULL/NULL to skip.  A unit whose grid exceeds its capacity sets
 *   status -4 and dumps nothing.
 *   marg (optional): doubles [n_units][2][marg_stride]: un-normalised marginals P_h1 / P_h2
 *   (models.py:277-285) indexed by position in the unit's h-axis (see tredgpu_grid_axis).
 */
int tredgpu_likelihood_grid(tredgpu_ctx* ctx, int mem, const tredgpu_unit_params* units,
                            int32_t n_units, int32_t hist_stride, const int32_t* full_cnt,
                            const int32_t* pref_cnt, const int32_t* rept_cnt,
                            const int32_t* global_lens, int64_t n_global_total,
                            const int32_t* target_lens, int64_t n_target_total,
                            tredgpu_call* calls, const int64_t* grid_off, double* grid_dump,
                            double* marg, int32_t marg_stride);

/*
 * tredgpu_likelihood_grid plus the sparse joint distribution the reference reports as P_h1h2
 * (models.py:279-285: P[(h1,h2)] = exp(ml - max); sparsify, :304-317: total = sum over the dict, keep v >= e^-10,
 * print v / total): per unit the triples {h1, h2, exp(ml - max)} of the distinct (h1,h2) pairs with
 * exp(ml - max) >= e^-10, in no particular order, and the sum over ALL distinct pairs.
 *   joint_off[n_units+1] (int64): caller-chosen capacities, in triples; joint[3 * joint_off[n_units]] doubles
 *   joint_n[n_units]: qualifying pairs of the unit -- when it exceeds the unit's capacity only the first
 *   `capacity` found were stored (ask again with more room, or use the grid dump)
 *   joint_total[n_units]: the normaliser
 * Saves the dense grid dump (48 bytes per pair) for callers that only format the JSON.
 */
int tredgpu_likelihood_grid_joint(tredgpu_ctx* ctx, int mem, const tredgpu_unit_params* units,
                                  int32_t n_units, int32_t hist_stride, const int32_t* full_cnt,
                                  const int32_t* pref_cnt, const int32_t* rept_cnt,
                                  const int32_t* global_lens, int64_t n_global_total,
                                  const int32_t* target_lens, int64_t n_target_total,
                                  tredgpu_call* calls, double* marg, int32_t marg_stride,
                                  const int64_t* joint_off, double* joint, int32_t* joint_n, double* joint_total);

/*
 * The fused path: SW + tagging -> histograms -> grid for a whole batch, nothing leaves the GPU
 * in between.  Arguments are the union of the three calls above; histograms are written to the
 * caller's buffers as well (they are the FR/PR/RR strings of the JSON, tred.py:254-256).
 */
int tredgpu_genotype_batch(tredgpu_ctx* ctx, int mem, const uint32_t* packed,
                           const int64_t* read_off, const int32_t* read_len, int64_t n_reads,
                           const int32_t* unit_read_off, const int32_t* unit_ladder,
                           const tredgpu_unit_params* units, int32_t n_units,
                           const tredgpu_sw_params* params, const int32_t* read_pair_id,
                           const int32_t* global_lens, int64_t n_global_total,
                           const int32_t* target_lens, int64_t n_target_total,
                           uint8_t* out_tag, int16_t* out_h, int16_t* out_score,
                           int32_t hist_stride, int32_t* full_cnt, int32_t* pref_cnt,
                           int32_t* rept_cnt, tredgpu_call* calls);

/*
 * The same for HOST memory with everything the product's writers print -- per-read tags, the calls, both marginals and the
 * sparse joint distribution (as tredgpu_likelihood_grid_joint returns them; a unit whose joint_n exceeds its room needs a
 * second call with more) -- in ONE call with ONE wait: tags, histograms and calls stay on the device between the stages.
 * rept_cnt (n_units x hist_stride, may be NULL) is the only histogram a caller still needs (the RR count).
 */
int tredgpu_genotype_batch_joint(tredgpu_ctx* ctx, const uint32_t* packed, const int64_t* read_off, const int32_t* read_len,
                                 int64_t n_reads, const int32_t* unit_read_off, const int32_t* unit_ladder,
                                 const tredgpu_unit_params* units, int32_t n_units, const tredgpu_sw_params* params,
                                 const int32_t* read_pair_id, const int32_t* global_lens, int64_t n_global_total,
                                 const int32_t* target_lens, int64_t n_target_total, uint8_t* out_tag, int16_t* out_h,
                                 int16_t* out_score, int32_t hist_stride, int32_t* rept_cnt, tredgpu_call* calls, double* marg,
                                 int32_t marg_stride, const int64_t* joint_off, double* joint, int32_t* joint_n,
                                 double* joint_total);

/*
 * KDE of the paired-end model alone (replaces gaussian_kde(global_lens).evaluate(arange(1000))
 * normalised to sum 1, models.py:428-435): pdf_out[n_units][1000]; units with fewer than 2
 * lengths or zero variance get status -2 in status_out.
 */
int tredgpu_pe_kde(tredgpu_ctx* ctx, int mem, const tredgpu_unit_params* units, int32_t n_units,
                   const int32_t* global_lens, int64_t n_global_total, double* pdf_out,
                   int32_t* status_out);

/* ---- measurement ------------------------------------------------------------------------------ */
/*
 * Every launch of the three main kernels is bracketed by HIP events on the context's stream.
 * tredgpu_get_timing synchronises the stream and returns, for kernel `which`
 * (TREDGPU_KERNEL_*), the number of launches since the last tredgpu_reset_timing and their summed
 * device time in milliseconds.  (No reference counterpart: the reference only prints wall time,
 * tredparse/tred.py:534-535.)
 */
#define TREDGPU_KERNEL_SW 0
#define TREDGPU_KERNEL_TALLY 1
#define TREDGPU_KERNEL_GRID 2          /* the four grid kernels of a call together ...    */
#define TREDGPU_KERNEL_GRID_PREPARE 3  /* ... and one by one: tables and axes per unit    */
#define TREDGPU_KERNEL_GRID_PAIRS 4    /*     the log-likelihood of every (h1, h2) pair   */
#define TREDGPU_KERNEL_GRID_REDUCE 5   /*     arg-max, marginals, CI, PP per unit         */
#define TREDGPU_KERNEL_GRID_KDE 6      /*     the paired-end KDEs (runs first)            */
int tredgpu_reset_timing(tredgpu_ctx* ctx);
int tredgpu_get_timing(tredgpu_ctx* ctx, int which, int64_t* launches, double* total_ms);
/*
 * Work counters of the SW kernel since the last tredgpu_reset_timing (one atomic set per wavefront):
 * out[0] trunk columns swept, out[1] continuation-pass columns swept, out[2] templates combined with the
 * continuation vectors, out[3] templates dropped by the exact score bounds, out[4] templates emitted from the
 * trunk state alone, out[5] wavefronts (quads of reads), out[6] read-columns = columns swept x reads in the
 * quad (empty slots of a partial quad not counted); out[7] reserved.  A column sweep occupies 4 read slots x
 * 16 lanes x R rows.  Lets bench.py report the cells really swept next to the brute-force count.
 */
int tredgpu_get_sw_counters(tredgpu_ctx* ctx, uint64_t out[8]);

/* ---- (4) BGZF block decoding for the read-selection front end --------------------------------------------------- */
/*
 * Raw DEFLATE (RFC 1951) decoding of many independent blocks of at most 64 KiB each in one launch: what htslib's
 * bgzf_read -> zlib inflate does block by block under the reference's pysam fetch / pileup calls
 * (tredparse/bam_parser.py:184-257, 316-369) -- two thirds of the host time of the from-BAM path.  The caller (the
 * host BAM layer, include/tredbam.h: tredbam_plan / tredbam_preload) strips the gzip framing, lays the payloads out in
 * the inflater's pinned staging buffer and checks CRC-32 / ISIZE on what comes back; this side only turns payloads
 * into bytes.  One inflater = one HIP stream + its staging; use one per host thread.  Waiting for a call sleeps
 * (blocking event), it does not spin: the host threads that wait are the ones whose cores the path is short of.
 *
 *   tredgpu_inflater_reserve  room for `comp_bytes` of payloads, `out_bytes` of output and n_blocks blocks; returns
 *                             the pinned host buffers: comp_host (fill), out_host (read after the call),
 *                             comp_off_host[n_blocks+1] / out_off_host[n_blocks+1] (fill: byte offsets into the two
 *                             buffers; every payload starts on a 4-byte boundary; out sizes <= 65536).  The pointers
 *                             stay valid until the next reserve that has to grow, or destroy.
 *   tredgpu_inflate_blocks    copies in, decodes, copies out (in slices: the copy-out of one slice runs beside the
 *                             decoding of the next), waits.  status[k]: 0, -1 invalid stream (or it runs
 *                             past its payload), -2 the stream ends before out_off[k+1]-out_off[k] bytes, -3 Huffman
 *                             codes that need more second-level table entries than the decoder holds (a guard: the
 *                             worst complete codes fit; the caller inflates such a block itself, like any other it
 *                             gets no bytes for).
 *                             Returns the number of blocks with a non-zero status, or <0.
 */
typedef struct tredgpu_inflater tredgpu_inflater;
int tredgpu_inflater_create(int device_id, tredgpu_inflater** out);
void tredgpu_inflater_destroy(tredgpu_inflater* inf);
const char* tredgpu_inflater_last_error(const tredgpu_inflater* inf);
int tredgpu_inflater_reserve(tredgpu_inflater* inf, int64_t comp_bytes, int64_t out_bytes, int32_t n_blocks,
                             uint8_t** comp_host, uint8_t** out_host, int64_t** comp_off_host, int64_t** out_off_host);
int tredgpu_inflate_blocks(tredgpu_inflater* inf, int32_t n_blocks, int32_t* status);
/*
 * The same, and the CRC-32 of every block's inflated bytes (0 where status != 0), computed on the device by the
 * wavefront that wrote them: the caller compares it with the BGZF trailer instead of walking the bytes again
 * (htslib checks the CRC inside bgzf_read, bgzf.c: the reference pays it in every pysam fetch).
 */
int tredgpu_inflate_blocks_crc(tredgpu_inflater* inf, int32_t n_blocks, int32_t* status, uint32_t* crc);
/*
 * Device time of the last call in milliseconds: from the first copy-in to the last copy-out, and the decode launches
 * alone (a call is cut into slices whose copies and launches overlap on two streams, so kernel_ms <= total_ms).
 */
int tredgpu_inflater_timing(tredgpu_inflater* inf, double* total_ms, double* kernel_ms);

/*
 * The pair-length walk where the blocks already are.  PEextractor (tredparse/bam_parser.py:316-369: the records of the
 * +-10 kb region of a locus, paired by query name, pair lengths from the soft-clipped ends, "target" pairs that span the
 * tract and "global" ones) reads 4 000 records per locus, ten times what the read selection and the depth need, and its
 * regions are most of the blocks of a sample.  tredgpu_inflate_walk decodes as tredgpu_inflate_blocks_crc does, then walks
 * the regions over the decoded blocks on the device (one region per wavefront) and brings back the pair lengths and,
 * per region, between which virtual offsets the records of the locus' own window lie; NO block is copied back.
 * tredgpu_inflater_fetch then copies the blocks the host asks for -- those between these offsets, a fifth of them -- to
 * their places in the pinned output.  include/tredbam.h (tredbam_plan_walks, tredbam_plan_blocks, tredbam_scan_pe) is the
 * host's half; the structs have the layouts of tredbam_walk_task / _chunk / _result.
 *   results[t].status  0: walked.  1 a block the walk needs is not among the sample's (or the file ends), 2 a block the
 *                      decoder rejected or whose CRC-32 is not blk_crc, 3 a record that makes no sense, 4 more query
 *                      names than the region's table holds (4 096, or 8 192 when a region of the call spans more
 *                      than 40 blocks) or more than 32 768 records, 5 a pair whose second read has no alignment end
 *                      (the reference dies there), 6 the pools are full, 7 two names under one hash: the host walks
 *                      that region itself (tredbam_scan_pe does).
 */
typedef struct tredgpu_walk_task {
    int32_t tid, start, end;        /* records of contig tid overlapping [start, end)                                     */
    int32_t tstart, tend, span;     /* a pair spans the tract when a.start < tstart and b.end > tend; tlen >= span: dropped */
    int32_t chunk_first, n_chunks;  /* its entries of chunks[]; n_chunks < 0: not walkable (status 1)                      */
    int32_t block_first, block_end; /* the blocks of the task's file among those of the call                              */
    int32_t win_lo, win_hi;         /* the window whose records' offsets are reported                                      */
} tredgpu_walk_task;
typedef struct tredgpu_walk_chunk { int32_t begin_block, begin_upos; uint64_t end_voffset; } tredgpu_walk_chunk;
typedef struct tredgpu_walk_result {
    int32_t status, n_global, n_target, n_window;
    int64_t global_first, target_first;     /* into the pools of the call                                                  */
    uint64_t win_vbeg, win_vend;            /* virtual offsets of the first window record / behind the last (0, 0: none)  */
} tredgpu_walk_result;
typedef struct tredgpu_walk_args {
    const int64_t* blk_coffset;             /* per block of the call: where it starts in its file                          */
    const int32_t* blk_clen;                /*                        its compressed length there                          */
    const uint32_t* blk_crc;                /*                        the CRC-32 its trailer promises                      */
    const tredgpu_walk_task* tasks;   int32_t n_tasks;
    const tredgpu_walk_chunk* chunks; int32_t n_chunks;
    tredgpu_walk_result* results;           /* out: n_tasks                                                                */
    int32_t* global_pool; int64_t cap_global;    /* out: the pair lengths (tasks take their room in any order)             */
    int32_t* target_pool; int64_t cap_target;
    int64_t n_global, n_target;             /* out: entries of the pools in use                                            */
    /* the walks over the alternative loci (tredbam_plan_alt_walks; n_alt_tasks == 0: none): in a task tstart is the contig
     * the MATE must lie on and [win_lo, win_hi] where (both ends included); results: status as above, and the virtual
     * offsets of the records that count, in file order (more than six: status 6).  need[k] != 0: block k holds one.   */
    const tredgpu_walk_task* alt_tasks;   int32_t n_alt_tasks;
    const tredgpu_walk_chunk* alt_chunks; int32_t n_alt_chunks;
    struct tredgpu_alt_result* alt_results; /* out: n_alt_tasks                                                            */
    uint8_t* need;                          /* out: n_blocks                                                               */
    /* the read selection where the records already are (section 5 below; select == NULL: none): one entry per task          */
    const struct tredgpu_select_task* select;
    struct tredgpu_select_result* selected; /* out: n_tasks                                                                */
} tredgpu_walk_args;
typedef struct tredgpu_alt_result { int32_t status, n; uint64_t vbeg[6]; } tredgpu_alt_result;
int tredgpu_inflate_walk(tredgpu_inflater* inf, int32_t n_blocks, int32_t* status, uint32_t* crc, tredgpu_walk_args* walk);
/* copies the blocks with need[k] != 0 of the last tredgpu_inflate_walk to the pinned output; returns the number of copies */
int tredgpu_inflater_fetch(tredgpu_inflater* inf, int32_t n_blocks, const uint8_t* need);
/*
 * The same without a pinned host copy of the whole output (45 MB per 30x sample and inflater, three inflaters per driver
 * process: the walk brings back a fifth of the blocks).  tredgpu_inflater_host_out(inf, 0): reserve no longer allocates
 * out_host (it returns NULL there; tredgpu_inflate_blocks[_crc] and tredgpu_inflater_fetch then fail with -2);
 * tredgpu_inflater_fetch_dense copies the wanted blocks -- and, as tredgpu_inflater_fetch does, the blocks between two
 * wanted ones that are fewer bytes than a copy costs -- one after the other into a pinned buffer of their own size:
 * block k lies at (*host)[dense_off[k] .. dense_off[k+1]) (empty when it was not copied; dense_off has n_blocks + 1
 * entries).  *host stays valid until the next fetch_dense of the inflater.  Returns the number of copies.
 */
int tredgpu_inflater_host_out(tredgpu_inflater* inf, int enabled);
int tredgpu_inflater_fetch_dense(tredgpu_inflater* inf, int32_t n_blocks, const uint8_t* need, uint8_t** host, int64_t* dense_off);
/* page-locked host memory the inflater holds at the moment, in bytes (its staging grows with the largest call it has seen) */
int64_t tredgpu_inflater_pinned_bytes(const tredgpu_inflater* inf);
/* of the last tredgpu_inflate_walk call: the regions whose records were listed by the serial chain (walk_chain_kernel) --
 * those the lane-parallel chain handed back: a guessed record start that was none, a record across the end of the planned
 * blocks, a block the decoder does not vouch for.  The results are the same either way (tests, diagnostics). */
int64_t tredgpu_inflater_walk_serial_regions(tredgpu_inflater* inf);
/* device time of the last call's walk launch in milliseconds */
int tredgpu_inflater_walk_ms(tredgpu_inflater* inf, double* walk_ms);

/* ---- (5) read selection, depth and 2-bit packing on the device ----------------------------------------------------------- */
/*
 * BamParser.parse's read selection (tredparse/bam_parser.py:199-243: of the window's records the unmapped ones -- placed at
 * their mate -- and those that start within one READLEN of the tract; then, from the alternative loci, the records whose
 * mate lies in the window), BamDepth.region_depth (:404-411: the pile-up of the window's records without truncation, summed)
 * and Aligner._DNA_to_int_mat (ssw_wrap.py:229-244) over the records tredgpu_inflate_walk has just listed: the inflated
 * blocks never leave the device -- what the host still reads of a sample are its calls, the tags and, for the JSON's
 * `details`, the selected reads' names and 4-bit sequences (~0.3 MB instead of ~7 MB of blocks per 30x sample).
 *
 *   tredgpu_select_task    one per task of the walk, in its order.  n_alt >= 0: a locus -- reads of the task's window
 *                          [win_lo, win_hi) that are unmapped or start within [pos_lo, pos_hi] are selected, in file order,
 *                          then the hits of the alternative regions alt_first .. alt_first + n_alt of the call's alt_tasks, region
 *                          by region (a region with n_chunks < 0 -- its contig is not in the file -- is skipped as the
 *                          reference's fetch skips it); depth_sum = sum over the window's records that are mapped, primary, not
 *                          QC-failed and not duplicates of (reference_end - reference_start).  n_alt < 0: a plain region (the
 *                          chrY windows of the sex inference, bam_parser.py:413-429): depth_sum only, nothing is selected; give
 *                          such a task span <= 0 and the pair walk leaves it alone.
 *   tredgpu_select_result  status 0, or: the pair walk's / an alternative region's status (1-7: the host scans the unit's sample
 *                          as before), 8 more than 4 096 reads, 9 a read beyond TREDGPU_MAX_READ_LEN.  n_words / seq4_bytes /
 *                          name_bytes: the room the unit's reads take in the packed layout (tredgpu_pack_reads), as 4-bit
 *                          sequences ((L + 1) / 2 bytes each) and as names (not terminated).
 */
typedef struct tredgpu_select_task { int32_t pos_lo, pos_hi, alt_first, n_alt; } tredgpu_select_task;
typedef struct tredgpu_select_result {
    int32_t status, n_reads, n_words, seq4_bytes, name_bytes, max_len;
    int64_t depth_sum;
} tredgpu_select_result;
#define TREDGPU_SELECT_CAP 4096            /* reads per unit the selection holds */
/*
 * tredgpu_genotype_batch_joint over reads that are still on the device: the units are tasks of inflaters' last
 * tredgpu_inflate_walk calls with a selection (segment s: n_units units, tasks task[.] of inflater inf, in batch order; the
 * inflaters' buffers must stay untouched until the call returns).  The reads are packed into the context's buffers by a
 * kernel, SW + tagging -> histograms -> grid run as in tredgpu_genotype_batch_joint, and ONE wait brings back what that call
 * returns plus the selected reads for the writers:
 *   unit_read_off[n_units + 1]: running sum of the units' n_reads; unit_word_off / unit_seq4_off / unit_name_off
 *   [n_units + 1] (int64): running sums of n_words / seq4_bytes / name_bytes (the caller has them from the select results);
 *   read_len[n_reads], seq4_off[n_reads + 1] + seq4[unit_seq4_off[n_units]], name_off[n_reads + 1] +
 *   names[unit_name_off[n_units]]: out, host.  params->max_read_len must name the longest selected read (max_len).
 */
typedef struct tredgpu_selected_units { tredgpu_inflater* inf; int32_t n_units; int32_t pad; const int32_t* task; } tredgpu_selected_units;
int tredgpu_genotype_selected(tredgpu_ctx* ctx, const tredgpu_selected_units* segs, int32_t n_segs, const int32_t* unit_read_off,
                              const int64_t* unit_word_off, const int64_t* unit_seq4_off, const int64_t* unit_name_off,
                              const int32_t* unit_ladder, const tredgpu_unit_params* units, int32_t n_units,
                              const tredgpu_sw_params* params, const int32_t* global_lens, int64_t n_global_total,
                              const int32_t* target_lens, int64_t n_target_total, uint8_t* out_tag, int16_t* out_h,
                              int16_t* out_score, int32_t hist_stride, int32_t* rept_cnt, tredgpu_call* calls, double* marg,
                              int32_t marg_stride, const int64_t* joint_off, double* joint, int32_t* joint_n, double* joint_total,
                              int32_t* read_len, int64_t* seq4_off, uint8_t* seq4, int64_t* name_off, char* names);

#ifdef __cplusplus
}
#endif
#endif /* TREDGPU_H */
