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

#ifdef __cplusplus
}
#endif
#endif
