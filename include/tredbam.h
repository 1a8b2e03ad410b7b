/* tredbam.h -- C ABI of libtredbam.so: the BAM file layer of the host read-selection front end.
 *
 * Replaces what the reference gets from pysam/htslib for this path (tredparse/bam_parser.py:22,432-436):
 *   samfile.fetch()                     all records in file order          bam_parser.py:384
 *   samfile.fetch(chr, start, end)      records overlapping [start, end)   bam_parser.py:206,226,333
 *   samfile.pileup(chr, start, end)     only ever summed to a depth        bam_parser.py:404-407
 *   samfile.getrname / references / lengths
 * Host-only C++ (zlib); no GPU involved.  tredparse_amd/bamio.py binds it through ctypes and keeps a pure-Python
 * implementation of the same layer; tests/test_host_frontend.py checks the two record for record.
 * CRAM is not supported (the reference hands .cram to htslib, bam_parser.py:435).
 */
#ifndef TREDBAM_H
#define TREDBAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tredbam tredbam;

/* Open a BAM file: reads the header; the .bai next to it (path + ".bai" or with .bam replaced) is loaded on
 * the first region query.  Returns 0, or <0 with the reason in tredbam_last_error(NULL). */
int tredbam_open(const char* path, tredbam** out);
void tredbam_close(tredbam* b);
/* message of the last failed call on b (b == NULL: of the last failed tredbam_open in this thread) */
const char* tredbam_last_error(const tredbam* b);

int32_t tredbam_n_ref(const tredbam* b);
const char* tredbam_ref_name(const tredbam* b, int32_t tid);   /* samfile.getrname(tid) */
int64_t tredbam_ref_len(const tredbam* b, int32_t tid);
int32_t tredbam_tid(const tredbam* b, const char* name);        /* -1: unknown contig */

/* One fetched record in the output buffer (little endian, 4-byte aligned, `size` bytes in all):
 *   tredbam_rec header, then l_name bytes of query name (NUL-terminated, as in the file), padding to 4,
 *   n_cigar uint32 (len << 4 | op), l_seq bytes of sequence as ASCII ("=ACMGRSVTWYHKDBN"), padding to 4.   */
typedef struct tredbam_rec {
    int32_t size;       /* bytes of this record incl. this header: the next record starts at +size */
    int32_t tid, pos;   /* reference id, 0-based leftmost position                                  */
    int32_t end;        /* pysam reference_end (one past the last aligned base); -1 without alignment */
    int32_t next_tid, next_pos, tlen;
    int32_t l_seq, n_cigar, l_name;
    uint16_t flag;
    uint8_t mapq, pad;
} tredbam_rec;

/* Records overlapping [start, end) on reference tid, found through the .bai (bins + linear index) like htslib;
 * placed-unmapped reads are returned at their mate's position.  tid < 0: all records in file order, at most
 * `limit` of them (limit <= 0: no limit).  *buf points into memory owned by b, valid until the next call on b.
 * Returns the number of records, or <0 on error. */
int64_t tredbam_fetch(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t limit, const uint8_t** buf,
                      int64_t* nbytes);

/* The read selection of BamParser.parse (bam_parser.py:206-214): of the records overlapping [start, end), the
 * unmapped ones (placed at their mate) and those with pos_lo <= pos <= pos_hi; same buffer layout as
 * tredbam_fetch. */
int64_t tredbam_fetch_reads(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t pos_lo, int64_t pos_hi,
                            const uint8_t** buf, int64_t* nbytes);

/* Sum over pileup columns of the number of reads covering them, for the reads that overlap [start, end):
 * every reference position such a read covers counts, also outside the region (pileup() without truncate,
 * bam_parser.py:404-407); unmapped / secondary / QC-fail / duplicate reads are skipped (htslib's default mask). */
int tredbam_pileup_depth_sum(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t* total);

/* PEextractor (bam_parser.py:316-369) over the records overlapping [start, end): reads that are paired, mapped
 * and not duplicates are grouped by query name in order of first appearance; for names seen at least twice the
 * first two records a, b must map +/-; tlen = b.reference_end (+ trailing soft clip) - a.reference_start
 * (- leading soft clip); pairs with tlen >= span are dropped; a pair with a.reference_start < tstart and
 * b.reference_end > tend spans the repeat (target_lens), any other goes to global_lens.  The counts are always
 * returned; lengths are written up to the given capacities (call again with larger arrays if a count exceeds
 * its capacity). */
int tredbam_pe_lengths(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t tstart, int64_t tend,
                       int32_t span, int32_t* global_lens, int64_t cap_global, int64_t* n_global,
                       int32_t* target_lens, int64_t cap_target, int64_t* n_target);

/* The block decoder behind the reader (csrc/inflate_block.h), exposed for tests: inflate the raw-deflate stream
 * in[0..n_in) whose output is exactly out_len bytes.  1 = decoded, 0 = declined (the reader would fall back to zlib),
 * < 0 bad arguments. */
int tredbam_inflate_raw(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t out_len);

/* The checksum every block is verified with after decoding (csrc/crc32_fold.h: carry-less-multiply folding with
 * zlib's crc32 as fallback), exposed for tests: CRC-32 of buf[0..len) continued from `crc` (0 to start).  A block
 * whose trailer does not match fails the call that needed it with -7, as htslib's bgzf_read_block does. */
uint32_t tredbam_crc32(uint32_t crc, const uint8_t* buf, int64_t len);

/* Largest l_seq among the first `first_n` records in file order (first_n <= 0: all): READLEN of a sample
 * (BamReadLen, bam_parser.py:372-391, which looks at 101 records). */
int tredbam_max_read_len(tredbam* b, int64_t first_n, int32_t* out);

/* ---- whole-sample scan: everything the GPU batch and the JSON need for a list of loci, in one call ------------
 * Replaces, per locus, BamDepth.region_depth (bam_parser.py:404-411), the read selection of BamParser.parse
 * (:196-243: window fetch, position filter, unmapped mates, ALT-locus mate rescue) and PEextractor (:316-369);
 * no per-record object ever reaches the host language.  Selected reads arrive already packed in libtredgpu's read
 * layout (include/tredgpu.h), in the order the reference would align them. */
typedef struct tredbam_site {
    int32_t tid;                  /* contig of the repeat in THIS file (tredbam_tid), < 0: not present           */
    int32_t repeat_start;         /* the locus table's repeat_start / repeat_end, used as the reference uses them */
    int32_t repeat_end;
    int32_t alt_first, n_alt;     /* this locus' entries in the alts[] array                                      */
} tredbam_site;

typedef struct tredbam_region { int32_t tid, start, end; } tredbam_region;   /* tid < 0: contig not in this file   */

typedef struct tredbam_scan_opts {
    int32_t readlen;     /* READLEN: reads must start within [repeat_start - readlen, repeat_end + readlen]        */
    int32_t pad;         /* SPAN = 1000: the fetch window is [repeat_start - pad, repeat_end + pad)                */
    int32_t flank;       /* FLANKMATCH = 9: a spanning pair starts before repeat_start - flank, ends after end + flank */
    int32_t pe_reach;    /* 10 x SPAN: pairs are collected within +- pe_reach of the tract                         */
    int32_t span;        /* pairs with tlen >= span are ignored                                                    */
    int32_t use_alts;    /* scan the alternative loci (off with --noalts or --useclippedreads)                     */
    int32_t want_depth, want_pe;
} tredbam_scan_opts;

#define TREDBAM_UNIT_NO_FETCH 1   /* unknown contig or no index: no reads (the reference logs and goes on)        */
#define TREDBAM_UNIT_FAILED 2     /* the file could not be read: the reference's exception drops the locus        */
#define TREDBAM_UNIT_NO_SEQ 4     /* a selected record holds no sequence (SEQ '*', l_seq 0): pysam's query_sequence is
                                   * None there and _parseReadSW's len(seq) raises (bam_parser.py:129-133): locus dropped */

typedef struct tredbam_unit {
    int32_t status;               /* TREDBAM_UNIT_* flags                                                           */
    int32_t n_reads;
    int64_t read_first;           /* first read of the unit in the pools                                            */
    int64_t depth_sum;            /* pileup depth sum over the window (divide by window length + 1)                 */
    int32_t depth_status;         /* != 0: the depth query failed (the reference falls back to depth 30)            */
    int32_t pe_status;            /* -9: a paired read without alignment end (TypeError in the reference)           */
    int32_t n_global, n_target;
    int64_t global_first, target_first;
} tredbam_unit;

typedef struct tredbam_pools {   /* memory owned by the handle, valid until its next scan                           */
    int64_t n_reads, n_words, n_global, n_target;
    const uint32_t* packed;       /* libtredgpu read records                                                        */
    const int64_t* word_off;      /* n_reads + 1                                                                    */
    const int32_t* read_len;
    const uint8_t* seq4;          /* the records' 4-bit sequences ("=ACMGRSVTWYHKDBN"), (L+1)/2 bytes each          */
    const int64_t* seq4_off;      /* n_reads + 1                                                                    */
    const char* names;            /* query names, not terminated                                                    */
    const int64_t* name_off;      /* n_reads + 1                                                                    */
    const int32_t* name_id;       /* index of the read's name among the distinct names of its unit                  */
    const int32_t* global_lens;
    const int32_t* target_lens;
} tredbam_pools;

/* Blocks inflated elsewhere -- on the GPU, include/tredgpu.h section 4 -- instead of by the scan itself (two thirds of a
 * scan's time is DEFLATE decoding):
 *   tredbam_plan       the BGZF blocks the region walks of tredbam_scan(sites, alts, opts) -- and of the caller's other
 *                      queries over the `extra` regions -- will read, from the index alone; returns their number, *comp_bytes = room their payloads take when each starts on a 4-byte
 *                      boundary, *out_bytes = their inflated size.
 *   tredbam_plan_fill  copies the payloads to comp + comp_off[k] from comp_base on and lays the outputs out from
 *                      out_base on; entry n of both offset arrays receives the end (= the next sample's bases).
 *   tredbam_preload    hands the inflated blocks in: block k at out + out_off[k] when status[k] == 0.  The handle keeps
 *                      pointers only; the next scan takes these blocks from there (CRC-32 checked at first use; a
 *                      block that fails it is dropped and inflated by the scan) and inflates whatever else it
 *                      needs itself, so a plan may miss blocks without harm.  Returns the
 *                      number of blocks taken.
 *   tredbam_preload_clear  forgets them (before the caller reuses the memory); reports how many block loads of the
 *                      scans since tredbam_preload were served from the preloaded set / were not. */
int64_t tredbam_plan(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                     const tredbam_scan_opts* o, const tredbam_region* extra, int32_t n_extra, int64_t* comp_bytes,
                     int64_t* out_bytes);
int tredbam_plan_fill(tredbam* b, uint8_t* comp, int64_t comp_base, int64_t out_base, int64_t* comp_off, int64_t* out_off);
int tredbam_preload(tredbam* b, const uint8_t* out, const int64_t* out_off, const int32_t* status);
/* The same when the decoder also delivers the CRC-32 of every block it wrote (tredgpu_inflate_blocks_crc): a block whose
 * checksum equals its BGZF trailer's is taken as verified -- the scan does not walk its bytes again --, one whose
 * checksum differs is not taken at all and is inflated and checked by the scan itself. */
int tredbam_preload_crc(tredbam* b, const uint8_t* out, const int64_t* out_off, const int32_t* status, const uint32_t* crc);
void tredbam_preload_clear(tredbam* b, int64_t* hits, int64_t* misses);

int tredbam_scan(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                 const tredbam_scan_opts* opts, tredbam_unit* units);
int tredbam_scan_pools(tredbam* b, tredbam_pools* pools);

/* ---- pair lengths computed where the blocks were inflated (the device: tredgpu_inflate_walk, tredgpu.h section 4) ----
 * PEextractor's walk (bam_parser.py:316-369) is most of a scan's time once the blocks arrive inflated (4 000 records per
 * locus, of which the read selection and the depth need a tenth), and its +-10 kb regions are two thirds of the blocks
 * that cross the bus.  The walk can run where the blocks already are:
 *   tredbam_plan_walks  after tredbam_plan: one task per site -- the region, the limits a pair is classified with, and the
 *                       region's merged index chunks as (index of the planned block the chunk starts in, offset in it,
 *                       end virtual offset).  n_chunks < 0: not walkable from the plan (unknown contig, no index, a chunk
 *                       that starts in a block the plan does not hold): the scan then computes that site itself.
 *                       Returns the number of chunks written, -3 when cap_chunks is too small.
 *   tredbam_plan_blocks per planned block (the plan is in file order): compressed offset, compressed length, the trailer's
 *                       CRC-32, and host[k] != 0 when the scan reads block k in any case (bit 0: alternative loci, bit 1:
 *                       extra regions).  With the walks done elsewhere only those and the blocks between a result's
 *                       win_vbeg and win_vend need to be handed to tredbam_preload.
 *   tredbam_scan_pe     tredbam_scan with the pair lengths of site i taken from pe[i] (status == 0: n_global / n_target
 *                       values from global_first / target_first of the two pools, in PEextractor's order); the scan
 *                       then reads only the records between pe[i].win_vbeg and win_vend -- those of the locus' window,
 *                       a tenth of the region's -- for the depth and the read selection.  Every other site is scanned
 *                       as tredbam_scan scans it.  Results are those of tredbam_scan. */
typedef struct tredbam_walk_task {
    int32_t tid, start, end;        /* records of contig tid overlapping [start, end)                                    */
    int32_t tstart, tend, span;     /* a pair spans the tract when a.start < tstart and b.end > tend; tlen >= span: dropped */
    int32_t chunk_first, n_chunks;  /* its entries of chunks[]                                                           */
    int32_t block_first, block_end; /* the sample's blocks among those of the call (tredbam_plan_walks: 0 .. plan size)  */
    int32_t win_lo, win_hi;         /* the scan's own window [win_lo, win_hi) inside the region: see win_vbeg / win_vend  */
} tredbam_walk_task;
typedef struct tredbam_walk_chunk { int32_t begin_block, begin_upos; uint64_t end_voffset; } tredbam_walk_chunk;
typedef struct tredbam_walk_result {
    int32_t status;                 /* 0: walked; anything else: not done here, the scan walks the site itself            */
    int32_t n_global, n_target;     /* pair lengths in PEextractor's order, from global_first / target_first of the pools  */
    int32_t n_window;               /* records of the region that overlap [win_lo, win_hi)                                 */
    int64_t global_first, target_first;
    uint64_t win_vbeg, win_vend;    /* virtual offsets of the first such record and behind the last one (0, 0: none)      */
} tredbam_walk_result;
int64_t tredbam_plan_walks(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_scan_opts* opts,
                           tredbam_walk_task* tasks, tredbam_walk_chunk* chunks, int64_t cap_chunks);
/* The same for the walks over the alternative loci (BamParser.parse's mate rescue, bam_parser.py:226-243: per locus ~50
 * small regions elsewhere in the genome, of whose records those count whose MATE lies in the locus' window): task
 * alt_first + k of site i is the region alts[alt_first + k] with tstart = the site's contig and [win_lo, win_hi] = the
 * window the mate must lie in (both ends included).  The walker returns per region the virtual offsets of the records
 * that count (tredbam_alt_result, at most six: more, or any status, and the scan walks the region itself), and
 * tredbam_scan_walked reads exactly those records -- each checked again to be what was promised.  In
 * tredbam_plan_blocks' flags bit 0 marks the blocks of these regions (needed on the host only for regions the walker
 * declined), bit 1 those of the caller's extra regions. */
typedef struct tredbam_alt_result { int32_t status, n; uint64_t vbeg[6]; } tredbam_alt_result;
int64_t tredbam_plan_alt_walks(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts, int32_t n_alts,
                               const tredbam_scan_opts* opts, tredbam_walk_task* tasks, tredbam_walk_chunk* chunks, int64_t cap_chunks);
int tredbam_scan_walked(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                        const tredbam_scan_opts* opts, const tredbam_walk_result* pe, const int32_t* pe_global,
                        const int32_t* pe_target, const tredbam_alt_result* alt_results, tredbam_unit* units);
int64_t tredbam_plan_blocks(tredbam* b, int64_t* coffset, int32_t* clen, uint32_t* crc, uint8_t* host);
/* Plain regions as tasks of the same walker (after tredbam_plan with the regions among its `extra`): task k = the records
 * overlapping regions[k], the window is the region, span 0 -- no pairs; for the chrY windows of the sex inference
 * (BamDepth.get_Y_depth, bam_parser.py:413-429), whose pile-up sums the device's read selection (include/tredgpu.h section 5)
 * returns.  Returns the number of chunks written, -3 when cap_chunks is too small; n_chunks < 0: not walkable. */
int64_t tredbam_plan_region_walks(tredbam* b, const tredbam_region* regions, int32_t n_regions, tredbam_walk_task* tasks,
                                  tredbam_walk_chunk* chunks, int64_t cap_chunks);
int tredbam_scan_pe(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                    const tredbam_scan_opts* opts, const tredbam_walk_result* pe, const int32_t* pe_global,
                    const int32_t* pe_target, tredbam_unit* units);
/* how many entries pe_global / pe_target of the next tredbam_scan_pe / tredbam_scan_walked hold: every pe[i] slice is then
 * checked against them (-2 on a slice outside; negative: unknown, the slices are trusted as the caller's) */
int tredbam_pe_pool_sizes(tredbam* b, int64_t n_global, int64_t n_target);

/* The JSON text of one locus' `details` list exactly as the driver prints it inside a sample's file (what
 * json.dumps(list, sort_keys=True, indent=4, separators=(',', ': ')) yields for the list at nesting depth 2:
 * elements {"h": int, "id": name, "seq": bases, "tag": "FULL" | "PREF" | "POST" | "REPT" | "HANG"}), written straight
 * from the pools of tredbam_scan_pools -- no per-read Python objects (tredparse/tred.py:118-121 stores the list, its
 * to_json :160-170 prints it).  reads[i] indexes the pools, tags[i] is the read's TREDGPU_TAG_* code (1..5), hs[i]
 * its repeat count.  Returns the number of bytes written; -3 when `cap` is too small; -1 when a name holds a byte
 * that json.dumps would not print as itself or as \" / \\ (control characters, non-ASCII: the caller's generic
 * encoder handles those); -2 on bad arguments. */
int64_t tredbam_details_json(const uint8_t* seq4, const int64_t* seq4_off, const int32_t* read_len, const char* names,
                             const int64_t* name_off, const int64_t* reads, const uint8_t* tags, const int32_t* hs,
                             int64_t n, char* out, int64_t cap);

/* The JSON text of one sparse distribution of the output -- `P_h1`, `P_h2` (keys "a") or `P_h1h2` (keys "a,b") of
 * tred.py's per-locus result (models.py:304-317 builds the dicts, tred.py:160-170 prints them) -- as
 * json.dumps(dict, sort_keys=True, indent=4, separators=(',', ': ')) prints it at nesting `depth`: keys sorted as
 * strings, values in Python's repr(float) (shortest digits that round-trip; exponent form below 1e-4 and from 1e16).
 * b == NULL: one-part keys.  Returns the bytes written; -3: `cap` too small; -1: not representable here (a value that
 * is not finite, a key listed twice) -- the caller's generic encoder then decides; -2: bad arguments. */
int64_t tredbam_sparse_json(const int32_t* a, const int32_t* b, const double* values, int64_t n, int32_t depth,
                            char* out, int64_t cap);
/* The two calls above for many items at once (a sample's 30 `details` lists and 90 distributions: one call each
 * instead of 120).  Item k covers entries off[k] .. off[k+1] of reads/tags/hs (a/b/values); two_part[k] != 0: keys
 * "a,b".  Item k's text is out[out_off[k] .. out_off[k+1]); status[k] = 0, or -1 where the single call would return -1
 * (empty text: the caller's generic encoder prints that item).  Return the bytes written, -3: cap too small, -2: bad
 * arguments. */
int64_t tredbam_sparse_json_many(const int32_t* a, const int32_t* b, const double* values, const int64_t* off,
                                 const uint8_t* two_part, int64_t n_items, int32_t depth, char* out, int64_t cap,
                                 int64_t* out_off, int8_t* status);
int64_t tredbam_details_json_many(const uint8_t* seq4, const int64_t* seq4_off, const int32_t* read_len, const char* names,
                                  const int64_t* name_off, const int64_t* reads, const uint8_t* tags, const int32_t* hs,
                                  const int64_t* off, int64_t n_items, char* out, int64_t cap, int64_t* out_off,
                                  int8_t* status);
/* Per slice pool[first[k] .. first[k] + count[k]) of a pair-length pool: mean, population standard deviation (0 for an
 * empty slice) and the 40-bin histogram of the values in [0, 1000] (bin = value / 25, the last bin closed) -- the numbers
 * the JSON's PEG / PET ("346+/-78bp") and P_PEG / P_PET ("0:0,25:0,...") strings are printed from
 * (tredparse/models.py:87-98), for all loci of a sample in one call.  hist: n x 40 ints.  0, or -2 on bad arguments. */
int tredbam_pair_stats(const int32_t* pool, const int64_t* first, const int32_t* count, int64_t n, double* mean,
                       double* sd, int32_t* hist);
/* repr(float) of one value into out (>= 32 bytes); returns the length, -1 for a value that is not finite (test hook) */
int tredbam_float_repr(double value, char* out);


/* ---- a sample's outputs written natively: <samplekey>.json and <samplekey>.tred.vcf.gz ---------------------------------
 * What the reference's run() tail and its writers do per sample in Python (tredparse/tred.py:251-275 the tredCalls keys,
 * :296-313 to_json: json.dumps(sort_keys=True, indent=4, separators=(',', ': ')), :316-374 to_vcf; bam_parser.py:174-182,
 * 248-287 the per-read bookkeeping; models.py:87-98 mean_std / histogram, :304-317 sparsify, :370-392 calc_label) from the
 * kernels' per-read and per-unit results and the scan's pools -- byte for byte the text tredparse_amd/tred.py's Python path
 * prints (tests/test_emit_native.py compares the two), without the interpreter lock: a driver process's Python work per
 * sample was what bounded the from-BAM rate. */
typedef struct tredbam_emit_locus {   /* one locus of the run's list, constant over the cohort                           */
    const char* name;                 /* "HD"                                                                            */
    const char* motif;                /* the repeat unit                                                                 */
    const char* chrom;                /* VCF CHROM                                                                       */
    const char* info;                 /* VCF INFO column without RPA ("END=...;MOTIF=...;...;VT=STR")                    */
    int32_t pos, ref_copy, period, cutoff_prerisk, cutoff_risk, is_expansion, is_recessive, in_vcf;
} tredbam_emit_locus;

typedef struct tredbam_emit_call {    /* = tredgpu_call (include/tredgpu.h)                                              */
    int32_t status, n_pairs, h1, h2, ci[4], run_pe, pad;
    double lik, pp;
} tredbam_emit_call;

typedef struct tredbam_emit_batch {   /* the arrays of one genotyped batch (host memory)                                 */
    const uint8_t* tag; const int16_t* h;            /* per read of the batch                                            */
    const int32_t* unit_read_off;                    /* reads of batch unit u: [unit_read_off[u], unit_read_off[u+1])    */
    const tredbam_emit_call* calls;                  /* per batch unit                                                   */
    const double* marg; int64_t marg_len;            /* [units][2][marg_len]: P_h1, P_h2 over alleles in repeat units    */
    const int64_t* joint_a; const int64_t* joint_b; const double* joint_v;   /* sparse joint entries, normalised, in units */
    const int64_t* joint_lo; const int32_t* joint_n; /* entries of unit u: [joint_lo[u], joint_lo[u] + joint_n[u])       */
    int32_t repeatpairs, pad;                        /* 0: reads of names tagged REPT twice are removed (--norepeatpairs) */
} tredbam_emit_batch;

typedef struct tredbam_emit_sample {
    const char* samplekey; const char* bam;          /* UTF-8                                                            */
    const char* gender; double ydepth;               /* inferredGender; depthY (< 0: printed as the integer -1)          */
    int32_t opened, readlen;                         /* opened == 0: only inferredGender / depthY are printed            */
    const uint8_t* seq4; const int64_t* seq4_off; const int32_t* read_len; const char* names; const int64_t* name_off;
    const int32_t* name_id; const int32_t* global_lens; const int32_t* target_lens;   /* the scan's pools                */
    const tredbam_unit* unit; const double* depth;   /* per locus of the list                                            */
    const int32_t* unit_index;                       /* per locus: its unit in the batch arrays, < 0: not genotyped      */
} tredbam_emit_sample;

typedef struct tredbam_emit_opts {
    const char* ref; const char* source; const char* filedate;   /* VCF header: ##reference, ##source prefix, ##fileDate */
    const char* vcf_meta;                            /* the ##INFO / ##FORMAT lines                                      */
    int32_t write_json, write_vcf, gzip_level, pad;
} tredbam_emit_opts;

/* Writes <samplekey>.json and <samplekey>.tred.vcf.gz into the current directory.  locus_status[n_loci]: 0 printed, 1 not
 * genotyped, < 0 the grid's status where the reference's grid raises (the locus is left out; the caller logs it).
 * json_text / json_cap: when json_text != NULL the JSON text is also copied there (for the echo on stdout), *json_len its
 * length (-(length) - 1 when json_cap is too small: nothing copied).  Returns 0; 1 when this sample needs the generic path (a name or key the fast printers do
 * not cover: non-ASCII read names, duplicate distribution keys, invalid UTF-8) -- nothing was written then; < 0 on errors
 * (-2 bad arguments, -5 a file could not be written: the message is in tredbam_emit_last_error()). */
int tredbam_emit_sample_files(const tredbam_emit_locus* loci, int32_t n_loci, const tredbam_emit_batch* batch,
                              const tredbam_emit_sample* sample, const tredbam_emit_opts* opts, int32_t* locus_status,
                              char* json_text, int64_t json_cap, int64_t* json_len);
const char* tredbam_emit_last_error(void);           /* of this thread                                                   */
/* numpy's pairwise float64 sum of a contiguous array (what `P.sum()` computes in models.py:309): test hook */
double tredbam_pairwise_sum(const double* a, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
