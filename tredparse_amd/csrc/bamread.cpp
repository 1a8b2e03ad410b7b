// libtredbam.so -- BGZF / BAM / BAI reader behind include/tredbam.h (host only, zlib).
//
// The file layer of the read-selection front end: what the reference takes from
// pysam/htslib at bam_parser.py:206,226,333,384,404-407.  Written against the SAM specification (sections
// 4.1 BGZF, 4.2 BAM, 5 indexing): virtual file offsets coffset << 16 | uoffset, the five-level binning scheme
// and the 16 kb linear index.  No third-party code.
#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "tredbam.h"

namespace {

thread_local std::string g_open_error;

uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

struct RefIndex {
    std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
};

}  // namespace

struct tredbam {
    std::string path, err;
    FILE* fp = nullptr;
    // current BGZF block
    int64_t block_coffset = -1, block_clen = 0;
    std::vector<uint8_t> block, cbuf;
    size_t upos = 0;
    // header
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::unordered_map<std::string, int32_t> tid_of;
    uint64_t first_record = 0;
    // index
    bool index_loaded = false;
    std::vector<RefIndex> index;
    // output of the last fetch
    std::vector<uint8_t> out;
    std::vector<uint8_t> rec;
};

namespace {

int fail(tredbam* b, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (b) b->err = buf; else g_open_error = buf;
    return code;
}

// Load the BGZF block that starts at compressed offset coffset.  Returns 1, 0 at end of file, <0 on error.
int load_block(tredbam* b, int64_t coffset) {
    b->block.clear();
    b->block_coffset = coffset;
    b->block_clen = 0;
    if (fseeko(b->fp, (off_t)coffset, SEEK_SET) != 0) return fail(b, -5, "seek to %lld failed", (long long)coffset);
    uint8_t hdr[18];
    const size_t got = fread(hdr, 1, sizeof hdr, b->fp);
    if (got < sizeof hdr) return 0;
    if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || hdr[3] != 4) return fail(b, -6, "not a BGZF block at %lld", (long long)coffset);
    const int xlen = le16(hdr + 10);
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, std::min<size_t>(6, extra.size()));
    if (xlen > 6 && fread(extra.data() + 6, 1, (size_t)xlen - 6, b->fp) != (size_t)xlen - 6) return fail(b, -6, "truncated BGZF block");
    int bsize = -1;
    for (int p = 0; p + 4 <= xlen;) {
        const int slen = le16(extra.data() + p + 2);
        if (extra[p] == 66 && extra[p + 1] == 67 && p + 6 <= xlen) bsize = le16(extra.data() + p + 4);
        p += 4 + slen;
    }
    if (bsize < 0) return fail(b, -6, "BGZF block without BC field");
    const int64_t clen = (int64_t)bsize + 1;
    const int64_t dlen = clen - 12 - xlen;   // deflate data + CRC32 + ISIZE
    if (dlen < 8) return fail(b, -6, "bad BGZF block size");
    b->cbuf.resize((size_t)dlen);
    if (fread(b->cbuf.data(), 1, (size_t)dlen, b->fp) != (size_t)dlen) return fail(b, -6, "truncated BGZF block");
    const uint32_t isize = le32(b->cbuf.data() + dlen - 4);
    b->block.resize(isize);
    if (isize > 0) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return fail(b, -7, "inflateInit2 failed");
        zs.next_in = b->cbuf.data();
        zs.avail_in = (uInt)(dlen - 8);
        zs.next_out = b->block.data();
        zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) return fail(b, -7, "inflate failed at %lld", (long long)coffset);
    }
    b->block_clen = clen;
    return 1;
}

int bg_seek(tredbam* b, uint64_t voffset) {
    const int64_t coffset = (int64_t)(voffset >> 16);
    if (coffset != b->block_coffset) {
        const int rc = load_block(b, coffset);
        if (rc < 0) return rc;
    }
    b->upos = (size_t)(voffset & 0xFFFF);
    return 0;
}

uint64_t bg_tell(const tredbam* b) { return ((uint64_t)b->block_coffset << 16) | (uint64_t)b->upos; }

// read n bytes; returns the number actually read (short at end of file), <0 on error
int64_t bg_read(tredbam* b, uint8_t* dst, int64_t n) {
    int64_t done = 0;
    while (n > 0) {
        if (b->upos >= b->block.size()) {
            const int rc = load_block(b, b->block_coffset + b->block_clen);
            if (rc < 0) return rc;
            b->upos = 0;
            if (rc == 0) break;            // end of file
            if (b->block.empty()) continue;  // empty block (e.g. the EOF marker): move on
        }
        const int64_t take = std::min<int64_t>(n, (int64_t)(b->block.size() - b->upos));
        memcpy(dst + done, b->block.data() + b->upos, (size_t)take);
        b->upos += (size_t)take;
        done += take;
        n -= take;
    }
    return done;
}

// next alignment record into b->rec (without its 4-byte block_size); 1 ok, 0 end of file, <0 error
int next_record(tredbam* b) {
    uint8_t head[4];
    const int64_t g = bg_read(b, head, 4);
    if (g < 0) return (int)g;
    if (g < 4) return 0;
    const int32_t size = (int32_t)le32(head);
    if (size < 32) return fail(b, -8, "bad alignment record size %d", size);
    b->rec.resize((size_t)size);
    const int64_t g2 = bg_read(b, b->rec.data(), size);
    if (g2 < 0) return (int)g2;
    if (g2 < size) return 0;
    return 1;
}

const bool CIGAR_REF[16] = {true, false, true, true, false, false, false, true, true};   // MIDNSHP=X

// append b->rec to b->out in the tredbam_rec layout; returns reference_end (-1: none) through *endp
int emit_record(tredbam* b, int32_t* endp, bool store) {
    const uint8_t* r = b->rec.data();
    const size_t size = b->rec.size();
    tredbam_rec h;
    h.tid = (int32_t)le32(r);
    h.pos = (int32_t)le32(r + 4);
    h.l_name = r[8];
    h.mapq = r[9];
    h.n_cigar = le16(r + 12);
    h.flag = le16(r + 14);
    h.l_seq = (int32_t)le32(r + 16);
    h.next_tid = (int32_t)le32(r + 20);
    h.next_pos = (int32_t)le32(r + 24);
    h.tlen = (int32_t)le32(r + 28);
    h.pad = 0;
    const size_t need = 32 + (size_t)h.l_name + 4 * (size_t)h.n_cigar + ((size_t)h.l_seq + 1) / 2;
    if (h.l_seq < 0 || need > size) return fail(b, -8, "alignment record shorter than its fields");
    const uint8_t* name = r + 32;
    const uint8_t* cig = name + h.l_name;
    const uint8_t* seq = cig + 4 * (size_t)h.n_cigar;
    int32_t end = -1;
    if (!(h.flag & 0x4) && h.n_cigar > 0) {
        int64_t e = h.pos;
        for (int k = 0; k < h.n_cigar; ++k) {
            const uint32_t c = le32(cig + 4 * k);
            if (CIGAR_REF[c & 15]) e += c >> 4;
        }
        end = (int32_t)e;
    }
    h.end = end;
    *endp = end;
    if (!store) return 0;
    const size_t name_pad = ((size_t)h.l_name + 3) & ~(size_t)3;
    const size_t seq_pad = ((size_t)h.l_seq + 3) & ~(size_t)3;
    h.size = (int32_t)(sizeof(tredbam_rec) + name_pad + 4 * (size_t)h.n_cigar + seq_pad);
    const size_t at = b->out.size();
    b->out.resize(at + (size_t)h.size, 0);
    uint8_t* o = b->out.data() + at;
    memcpy(o, &h, sizeof h);
    o += sizeof h;
    memcpy(o, name, h.l_name);
    o += name_pad;
    memcpy(o, cig, 4 * (size_t)h.n_cigar);   // the file is little endian, as is every host this builds for
    o += 4 * (size_t)h.n_cigar;
    static const char SEQ[] = "=ACMGRSVTWYHKDBN";
    for (int32_t i = 0; i < h.l_seq; ++i) {
        const uint8_t v = seq[i >> 1];
        o[i] = (uint8_t)SEQ[(i & 1) ? (v & 15) : (v >> 4)];
    }
    return 0;
}

int load_index(tredbam* b) {
    if (b->index_loaded) return 0;
    std::string cand[2] = {b->path + ".bai", b->path};
    const size_t dot = cand[1].rfind('.');
    if (dot != std::string::npos) cand[1] = cand[1].substr(0, dot);
    cand[1] += ".bai";
    FILE* f = nullptr;
    for (const std::string& c : cand)
        if ((f = fopen(c.c_str(), "rb")) != nullptr) break;
    if (!f) return fail(b, -4, "no .bai index next to %s", b->path.c_str());
    std::vector<uint8_t> d;
    uint8_t tmp[65536];
    size_t g;
    while ((g = fread(tmp, 1, sizeof tmp, f)) > 0) d.insert(d.end(), tmp, tmp + g);
    fclose(f);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return fail(b, -4, "bad BAI magic");
    const int32_t n_ref = (int32_t)le32(d.data() + 4);
    size_t p = 8;
    b->index.assign((size_t)std::max(n_ref, 0), RefIndex());
    for (int32_t t = 0; t < n_ref; ++t) {
        if (p + 4 > d.size()) return fail(b, -4, "truncated BAI");
        const int32_t n_bin = (int32_t)le32(d.data() + p); p += 4;
        for (int32_t k = 0; k < n_bin; ++k) {
            if (p + 8 > d.size()) return fail(b, -4, "truncated BAI");
            const uint32_t bin = le32(d.data() + p);
            const int32_t n_chunk = (int32_t)le32(d.data() + p + 4); p += 8;
            if (n_chunk < 0 || p + 16 * (size_t)n_chunk > d.size()) return fail(b, -4, "truncated BAI");
            auto& v = b->index[t].bins[bin];
            for (int32_t c = 0; c < n_chunk; ++c, p += 16) v.emplace_back(le64(d.data() + p), le64(d.data() + p + 8));
        }
        if (p + 4 > d.size()) return fail(b, -4, "truncated BAI");
        const int32_t n_intv = (int32_t)le32(d.data() + p); p += 4;
        if (n_intv < 0 || p + 8 * (size_t)n_intv > d.size()) return fail(b, -4, "truncated BAI");
        b->index[t].linear.resize((size_t)n_intv);
        for (int32_t k = 0; k < n_intv; ++k, p += 8) b->index[t].linear[k] = le64(d.data() + p);
    }
    b->index_loaded = true;
    return 0;
}

// Region walk shared by fetch and the depth sum.  visit(end) is called for every record overlapping the region
// after emit_record() has parsed it (and stored it when `store`).
template <typename F>
int64_t walk_region(tredbam* b, int32_t tid, int64_t start, int64_t end, bool store, F visit) {
    if (tid < 0 || tid >= (int32_t)b->ref_names.size()) return fail(b, -2, "invalid contig id %d", tid);
    start = std::max<int64_t>(0, start);
    if (end < 0) end = b->ref_lens[tid];
    if (start > end) return fail(b, -2, "invalid coordinates: start > end");
    int rc = load_index(b);
    if (rc) return rc;
    int64_t n = 0;
    if (tid >= (int32_t)b->index.size()) return 0;
    const RefIndex& ix = b->index[tid];
    uint64_t min_off = 0;
    if (!ix.linear.empty()) min_off = ix.linear[std::min<size_t>((size_t)(start >> 14), ix.linear.size() - 1)];
    // reg2bins of the 5-level scheme over [start, max(end, start + 1))
    const int64_t e1 = std::max(end, start + 1) - 1;
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    auto add_bin = [&](uint32_t bin) {
        auto it = ix.bins.find(bin);
        if (it == ix.bins.end()) return;
        for (const auto& ch : it->second)
            if (ch.second > min_off) chunks.emplace_back(std::max(ch.first, min_off), ch.second);
    };
    add_bin(0);
    const int shifts[5] = {26, 23, 20, 17, 14};
    const uint32_t bases[5] = {1, 9, 73, 585, 4681};
    for (int l = 0; l < 5; ++l)
        for (int64_t k = start >> shifts[l]; k <= e1 >> shifts[l]; ++k) add_bin(bases[l] + (uint32_t)k);
    std::sort(chunks.begin(), chunks.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (const auto& ch : chunks) {
        if (!merged.empty() && ch.first <= merged.back().second) merged.back().second = std::max(merged.back().second, ch.second);
        else merged.push_back(ch);
    }
    for (const auto& ch : merged) {
        if ((rc = bg_seek(b, ch.first)) < 0) return rc;
        while (bg_tell(b) < ch.second) {
            rc = next_record(b);
            if (rc < 0) return rc;
            if (rc == 0) break;
            const int32_t rtid = (int32_t)le32(b->rec.data());
            const int32_t rpos = (int32_t)le32(b->rec.data() + 4);
            if (rtid != tid || rpos >= end) {
                if (rtid > tid || (rtid == tid && rpos >= end)) break;
                continue;
            }
            // overlap test first (cheap pass without storing), then store
            const size_t mark = b->out.size();
            int32_t rend;
            if ((rc = emit_record(b, &rend, store)) < 0) return rc;
            int64_t e = rend;
            if (rend < 0 || rend <= rpos) e = (int64_t)rpos + 1;   // placed-unmapped / zero length: one base (bam_endpos)
            if (e > start && visit(rend, rpos, le16(b->rec.data() + 14), b->rec.data())) ++n;
            else b->out.resize(mark);
        }
    }
    return n;
}

}  // namespace

extern "C" {

int tredbam_open(const char* path, tredbam** out) {
    if (!out) return -2;
    *out = nullptr;
    if (!path) return fail(nullptr, -2, "path is NULL");
    const std::string p(path);
    if (p.size() >= 5 && p.compare(p.size() - 5, 5, ".cram") == 0) return fail(nullptr, -3, "CRAM is not supported by this front end");
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(nullptr, -4, "file `%s` not found", path);
    tredbam* b = new tredbam();
    b->path = p;
    b->fp = fp;
    auto bail = [&](int code) {
        g_open_error = b->err;
        fclose(b->fp);
        delete b;
        return code;
    };
    int rc = bg_seek(b, 0);
    if (rc < 0) return bail(rc);
    uint8_t w[8];
    if (bg_read(b, w, 4) != 4 || memcmp(w, "BAM\1", 4) != 0) { b->err = "not a BAM file: " + p; return bail(-6); }
    if (bg_read(b, w, 4) != 4) { b->err = "truncated BAM header"; return bail(-6); }
    const int32_t l_text = (int32_t)le32(w);
    std::vector<uint8_t> text((size_t)std::max(l_text, 0));
    if (bg_read(b, text.data(), l_text) != l_text || bg_read(b, w, 4) != 4) { b->err = "truncated BAM header"; return bail(-6); }
    const int32_t n_ref = (int32_t)le32(w);
    for (int32_t t = 0; t < n_ref; ++t) {
        if (bg_read(b, w, 4) != 4) { b->err = "truncated BAM header"; return bail(-6); }
        const int32_t l_name = (int32_t)le32(w);
        std::vector<uint8_t> nm((size_t)std::max(l_name, 1));
        if (bg_read(b, nm.data(), l_name) != l_name || bg_read(b, w, 4) != 4) { b->err = "truncated BAM header"; return bail(-6); }
        b->ref_names.emplace_back((const char*)nm.data(), (size_t)std::max(l_name - 1, 0));
        b->ref_lens.push_back((int32_t)le32(w));
        b->tid_of[b->ref_names.back()] = t;
    }
    b->first_record = bg_tell(b);
    *out = b;
    return 0;
}

void tredbam_close(tredbam* b) {
    if (!b) return;
    if (b->fp) fclose(b->fp);
    delete b;
}

const char* tredbam_last_error(const tredbam* b) { return b ? b->err.c_str() : g_open_error.c_str(); }

int32_t tredbam_n_ref(const tredbam* b) { return b ? (int32_t)b->ref_names.size() : 0; }

const char* tredbam_ref_name(const tredbam* b, int32_t tid) {
    if (!b || tid < 0 || tid >= (int32_t)b->ref_names.size()) return nullptr;
    return b->ref_names[tid].c_str();
}

int64_t tredbam_ref_len(const tredbam* b, int32_t tid) {
    if (!b || tid < 0 || tid >= (int32_t)b->ref_lens.size()) return -1;
    return b->ref_lens[tid];
}

int32_t tredbam_tid(const tredbam* b, const char* name) {
    if (!b || !name) return -1;
    auto it = b->tid_of.find(name);
    return it == b->tid_of.end() ? -1 : it->second;
}

int64_t tredbam_fetch(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t limit, const uint8_t** buf,
                      int64_t* nbytes) {
    if (!b || !buf || !nbytes) return -2;
    b->out.clear();
    int64_t n = 0;
    if (tid < 0) {   // whole file, in order
        int rc = bg_seek(b, b->first_record);
        if (rc < 0) return rc;
        while (limit <= 0 || n < limit) {
            rc = next_record(b);
            if (rc < 0) return rc;
            if (rc == 0) break;
            int32_t rend;
            if ((rc = emit_record(b, &rend, true)) < 0) return rc;
            ++n;
        }
    } else {
        n = walk_region(b, tid, start, end, true, [](int32_t, int32_t, uint16_t, const uint8_t*) { return true; });
        if (n < 0) return n;
    }
    *buf = b->out.data();
    *nbytes = (int64_t)b->out.size();
    return n;
}

int64_t tredbam_fetch_reads(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t pos_lo, int64_t pos_hi,
                            const uint8_t** buf, int64_t* nbytes) {
    if (!b || !buf || !nbytes) return -2;
    b->out.clear();
    const int64_t n = walk_region(b, tid, start, end, true, [&](int32_t, int32_t rpos, uint16_t flag, const uint8_t*) {
        return (flag & 0x4) != 0 || (rpos >= pos_lo && rpos <= pos_hi);
    });
    if (n < 0) return n;
    *buf = b->out.data();
    *nbytes = (int64_t)b->out.size();
    return n;
}

int tredbam_pileup_depth_sum(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t* total) {
    if (!b || !total) return -2;
    b->out.clear();
    int64_t sum = 0;
    const int64_t n = walk_region(b, tid, start, end, false, [&](int32_t rend, int32_t rpos, uint16_t flag, const uint8_t*) {
        if (flag & (0x4 | 0x100 | 0x200 | 0x400)) return true;   // unmapped, secondary, QC fail, duplicate
        if (rend >= 0) sum += (int64_t)rend - rpos;
        return true;
    });
    if (n < 0) return (int)n;
    *total = sum;
    return 0;
}

int tredbam_pe_lengths(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t tstart, int64_t tend,
                       int32_t span, int32_t* global_lens, int64_t cap_global, int64_t* n_global,
                       int32_t* target_lens, int64_t cap_target, int64_t* n_target) {
    if (!b || !n_global || !n_target) return -2;
    b->out.clear();
    struct Mate { int32_t pos, end, l_seq, lead_clip, trail_clip; bool reverse; };
    struct Pair { int n; Mate m[2]; };
    std::unordered_map<std::string, size_t> slot;
    std::vector<Pair> pairs;   // in order of first appearance (the reference walks a dict in that order)
    const int64_t n = walk_region(b, tid, start, end, false, [&](int32_t rend, int32_t rpos, uint16_t flag, const uint8_t* r) {
        if (!(flag & 0x1) || (flag & 0x4) || (flag & 0x400)) return true;   // paired, mapped, not a duplicate
        const int l_name = r[8];
        const int n_cigar = le16(r + 12);
        std::string name((const char*)r + 32, (size_t)std::max(l_name - 1, 0));
        auto it = slot.find(name);
        if (it == slot.end()) {
            it = slot.emplace(std::move(name), pairs.size()).first;
            pairs.push_back(Pair{0, {}});
        }
        Pair& p = pairs[it->second];
        if (p.n < 2) {
            Mate& m = p.m[p.n];
            m.pos = rpos;
            m.end = rend;
            m.l_seq = (int32_t)le32(r + 16);
            m.reverse = (flag & 0x10) != 0;
            const uint8_t* cig = r + 32 + l_name;
            int32_t lead = 0, trail = 0;
            for (int k = 0; k < n_cigar; ++k) {            // query_alignment_start: leading soft clips
                const uint32_t c = le32(cig + 4 * k);
                if ((c & 15) == 4) lead += (int32_t)(c >> 4);
                else if ((c & 15) == 5) continue;
                else break;
            }
            for (int k = n_cigar - 1; k >= 0; --k) {       // query_length - query_alignment_end
                const uint32_t c = le32(cig + 4 * k);
                if ((c & 15) == 4) trail += (int32_t)(c >> 4);
                else if ((c & 15) == 5) continue;
                else break;
            }
            m.lead_clip = lead;
            m.trail_clip = trail;
        }
        ++p.n;
        return true;
    });
    if (n < 0) return (int)n;
    int64_t ng = 0, nt = 0;
    for (const Pair& p : pairs) {
        if (p.n < 2) continue;
        const Mate &a = p.m[0], &bb = p.m[1];
        if (a.reverse || !bb.reverse) continue;            // mapped in +, - orientation
        if (bb.end < 0) return fail(b, -9, "paired read without an alignment end in the window");
        const int64_t tlen = ((int64_t)bb.end + bb.trail_clip) - ((int64_t)a.pos - a.lead_clip);
        if (tlen >= span) continue;
        if (a.pos < tstart && bb.end > tend) {
            if (target_lens && nt < cap_target) target_lens[nt] = (int32_t)tlen;
            ++nt;
        } else {
            if (global_lens && ng < cap_global) global_lens[ng] = (int32_t)tlen;
            ++ng;
        }
    }
    *n_global = ng;
    *n_target = nt;
    return 0;
}

}  // extern "C"
