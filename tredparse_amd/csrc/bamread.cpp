// libtredbam.so -- BGZF / BAM / BAI reader behind include/tredbam.h (host only, zlib).
//
// The file layer of the read-selection front end: what the reference takes from
// pysam/htslib at bam_parser.py:206,226,333,384,404-407.  Written against the SAM specification (sections
// 4.1 BGZF, 4.2 BAM, 5 indexing): virtual file offsets coffset << 16 | uoffset, the five-level binning scheme
// and the 16 kb linear index.  No third-party code.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "crc32_fold.h"
#include "inflate_block.h"
#include "tredbam.h"

namespace {

thread_local std::string g_open_error;

uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// One contig of the .bai: where its bins and its linear index lie in the file's bytes (tredbam::bai).  The bins are parsed
// into the map when the contig is first queried, the linear index is read in place: a whole-genome .bai is ~8 MB of
// which a sample's queries touch a few contigs' bins and a handful of linear entries -- copying all of it into maps and
// vectors per sample was a millisecond of every synthetic sample's plan (1.1 MB .bai) and several of a real one's.
struct RefIndex {
    std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    bool bins_parsed = false;
    size_t bins_at = 0;             // offset of the contig's first bin record
    int32_t n_bin = 0;
    const uint8_t* lin = nullptr;   // n_lin little-endian 64-bit entries
    size_t n_lin = 0;
    uint64_t linear(size_t k) const { return le64(lin + 8 * k); }
};

}  // namespace

// Buffers of inflated blocks are recycled per thread: a sample inflates ~550 blocks of 64 KiB, and handing 35 MB back
// to the allocator at every close meant fresh pages (faults, zeroing) for every sample a scan thread takes next.
namespace tredbam_pool {
constexpr size_t BLOCK_BYTES = 65536 + tredbam_inflate::SLACK;
constexpr size_t KEEP = 640;                     // buffers kept per thread (~42 MB)
struct Pool {
    std::vector<uint8_t*> free;
    ~Pool() { for (uint8_t* p : free) delete[] p; }
};
inline Pool& pool() { static thread_local Pool p; return p; }
inline uint8_t* take() {
    Pool& p = pool();
    if (p.free.empty()) return new uint8_t[BLOCK_BYTES];
    uint8_t* b = p.free.back();
    p.free.pop_back();
    return b;
}
struct Give {
    void operator()(uint8_t* b) const {
        Pool& p = pool();
        if (p.free.size() < KEEP) p.free.push_back(b); else delete[] b;
    }
};
using Buffer = std::unique_ptr<uint8_t[], Give>;
}  // namespace tredbam_pool

struct tredbam {
    std::string path, err;
    FILE* fp = nullptr;
    // the file: mapped read-only when the system allows it (a block's compressed bytes are then decoded where they
    // lie), else read through `fp` into `cbuf`
    const uint8_t* map = nullptr;
    size_t map_size = 0;
    // current BGZF block: `block` points into the cache entry of `block_coffset` (valid until that entry is evicted,
    // i.e. at least until the next load_block), block_size inflated bytes
    int64_t block_coffset = -1, block_clen = 0;
    const uint8_t* block = nullptr;
    size_t block_size = 0;
    std::vector<uint8_t> cbuf;
    size_t upos = 0;
    // inflated blocks seen recently (compressed offset -> data, compressed length): the three queries of a locus
    // (depth, reads, pairs) and the alternative-locus queries of neighbouring loci walk the same blocks again.
    // An entry owns its buffer (no zero fill, never copied); the current block is used in place.
    struct Cached { tredbam_pool::Buffer data; size_t size; int64_t clen; };
    std::unordered_map<int64_t, Cached> cache;
    std::deque<int64_t> cache_order;
    static constexpr size_t CACHE_BLOCKS = 512;   // <= 32 MiB per open file
    // blocks inflated elsewhere (tredbam_plan -> the GPU's batch decoder -> tredbam_preload): compressed offset -> the
    // caller's bytes; load_block takes them from here (CRC checked at first use), anything else is inflated as usual
    struct Planned { int64_t coffset, payload_off; int32_t payload_len; int64_t clen; uint32_t crc, isize; uint8_t host; };
    struct Preloaded { const uint8_t* data; uint32_t size; int64_t clen; uint32_t crc; bool checked; };
    std::vector<Planned> plan;
    std::unordered_map<int64_t, Preloaded> preloaded;
    int64_t preload_hits = 0, preload_misses = 0;
    int64_t pe_pool_global = -1, pe_pool_target = -1;   // lengths of the pools tredbam_scan_pe / _walked index (tredbam_pe_pool_sizes)
    tredbam_inflate::Tables inflate_tables;        // decoding tables of the block decoder (inflate_block.h)
    // header
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::unordered_map<std::string, int32_t> tid_of;
    uint64_t first_record = 0;
    // index
    bool index_loaded = false;
    std::vector<uint8_t> bai;            // the .bai file's bytes when it could not be mapped
    const uint8_t* bai_p = nullptr;      // the .bai file's bytes (RefIndex points into them): mapped read-only, so that only the
    size_t bai_n = 0;                    //   pages a sample's queries touch are ever looked at (1.1 MB - 8 MB per file), or `bai`
    bool bai_mapped = false;
    std::vector<RefIndex> index;
    // output of the last fetch
    std::vector<uint8_t> out;
    std::vector<uint8_t> rec;            // a record that straddles blocks is assembled here
    const uint8_t* recp = nullptr;       // the current record (into `block` or `rec`), rec_size bytes
    size_t rec_size = 0;
    // pools of the last tredbam_scan (include/tredbam.h)
    std::vector<uint32_t> sc_packed;
    std::vector<int64_t> sc_word_off, sc_seq4_off, sc_name_off;
    std::vector<int32_t> sc_read_len, sc_name_id, sc_global, sc_target;
    std::vector<uint8_t> sc_seq4;
    std::vector<char> sc_names;
    bool sc_noseq = false;               // pool_read met a record without a sequence since the flag was last cleared
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
// n bytes of the file at `off`: in the mapping, or read into b->cbuf; nullptr when the file ends before
const uint8_t* file_bytes(tredbam* b, int64_t off, size_t n) {
    if (b->map) return (off >= 0 && (uint64_t)off + n <= b->map_size) ? b->map + off : nullptr;
    b->cbuf.resize(n);
    if (fseeko(b->fp, (off_t)off, SEEK_SET) != 0) return nullptr;
    return fread(b->cbuf.data(), 1, n, b->fp) == n ? b->cbuf.data() : nullptr;
}

// The block decoder, compiled a second time for CPUs with BMI2 / AVX2 (variable shifts without the count register,
// bzhi for the extra bits, wider copies: ~12 % faster there); the loader picks the clone the CPU supports.
__attribute__((target_clones("arch=haswell", "default")))
bool inflate_dispatch(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, tredbam_inflate::Tables& T) {
    return tredbam_inflate::inflate_block(in, in_len, out, out_len, T);
}

// The frame of the BGZF block at compressed offset coffset: where its deflate payload lies, how long the block is, and
// the trailer.  Returns 1, 0 at end of file, <0 on error.  *comp: the payload + trailer bytes (valid until the next
// file_bytes call when the file is not mapped).
struct BlockFrame { int64_t clen, payload_off, dlen; uint32_t crc, isize; };
int block_frame(tredbam* b, int64_t coffset, BlockFrame& f, const uint8_t** comp) {
    uint8_t hdr[18];
    {
        const uint8_t* h = file_bytes(b, coffset, sizeof hdr);
        if (!h) return 0;                              // end of file
        memcpy(hdr, h, sizeof hdr);
    }
    if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || hdr[3] != 4) return fail(b, -6, "not a BGZF block at %lld", (long long)coffset);
    const int xlen = le16(hdr + 10);
    int bsize = -1;
    {
        const uint8_t* extra = file_bytes(b, coffset + 12, (size_t)xlen);
        if (!extra) return fail(b, -6, "truncated BGZF block");
        for (int p = 0; p + 4 <= xlen;) {
            const int slen = le16(extra + p + 2);
            if (extra[p] == 66 && extra[p + 1] == 67 && p + 6 <= xlen) bsize = le16(extra + p + 4);
            p += 4 + slen;
        }
    }
    if (bsize < 0) return fail(b, -6, "BGZF block without BC field");
    f.clen = (int64_t)bsize + 1;
    f.dlen = f.clen - 12 - xlen;   // deflate data + CRC32 + ISIZE
    if (f.dlen < 8) return fail(b, -6, "bad BGZF block size");
    f.payload_off = coffset + 12 + xlen;
    const uint8_t* c = file_bytes(b, f.payload_off, (size_t)f.dlen);
    if (!c) return fail(b, -6, "truncated BGZF block");
    f.crc = le32(c + f.dlen - 8);
    f.isize = le32(c + f.dlen - 4);
    // a BGZF block holds at most 64 KiB (SAM spec 4.1): a larger ISIZE is a damaged trailer, not a 4 GiB allocation
    if (f.isize > 65536) return fail(b, -6, "BGZF block at %lld claims %u bytes", (long long)coffset, f.isize);
    *comp = c;
    return 1;
}

int load_block(tredbam* b, int64_t coffset) {
    b->block_coffset = coffset;
    {
        const auto hit = b->cache.find(coffset);
        if (hit != b->cache.end()) {
            b->block = hit->second.data.get();
            b->block_size = hit->second.size;
            b->block_clen = hit->second.clen;
            return 1;
        }
    }
    b->block = nullptr;
    b->block_size = 0;
    b->block_clen = 0;
    if (!b->preloaded.empty()) {
        const auto pre = b->preloaded.find(coffset);
        if (pre != b->preloaded.end()) {
            tredbam::Preloaded& p = pre->second;
            // the block's CRC-32, as for a block inflated here (already compared when the decoder delivered one).  A
            // preloaded block that fails it is dropped and inflated here like a block the plan missed: only what the
            // file itself holds can fail a scan
            if (p.checked || tredbam_crc::crc32(0, p.data, p.size) == p.crc) {
                p.checked = true;
                b->block = p.data;
                b->block_size = p.size;
                b->block_clen = p.clen;
                ++b->preload_hits;
                return 1;
            }
            b->preloaded.erase(pre);
        }
        ++b->preload_misses;
    }
    BlockFrame fr = {};
    const uint8_t* comp = nullptr;
    {
        const int rc = block_frame(b, coffset, fr, &comp);
        if (rc <= 0) return rc;
    }
    const int64_t clen = fr.clen, dlen = fr.dlen;
    const uint32_t isize = fr.isize;
    tredbam_pool::Buffer data(tredbam_pool::take());     // (isize <= 65536: every block fits a pooled buffer)
    // own whole-block decoder first (1.3-1.6x zlib's speed on BAM data); zlib decides whenever it declines
    const bool done = isize > 0 && inflate_dispatch(comp, (size_t)(dlen - 8), data.get(), isize, b->inflate_tables);
    if (isize > 0 && !done) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return fail(b, -7, "inflateInit2 failed");
        zs.next_in = const_cast<uint8_t*>(comp);
        zs.avail_in = (uInt)(dlen - 8);
        zs.next_out = data.get();
        zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) return fail(b, -7, "inflate failed at %lld", (long long)coffset);
    }
    // the block's CRC-32 trailer, after either decoder (htslib checks it too: a damaged block that still inflates
    // to ISIZE bytes must be an error, not a genotype)
    if (tredbam_crc::crc32(0, data.get(), isize) != le32(comp + dlen - 8))
        return fail(b, -7, "CRC mismatch in the BGZF block at %lld", (long long)coffset);
    b->block_clen = clen;
    if (b->cache.size() >= tredbam::CACHE_BLOCKS) {
        b->cache.erase(b->cache_order.front());
        b->cache_order.pop_front();
    }
    const auto at = b->cache.emplace(coffset, tredbam::Cached{std::move(data), isize, clen}).first;
    b->cache_order.push_back(coffset);
    b->block = at->second.data.get();
    b->block_size = isize;
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

// A position at (or, for a 64 KiB block, past the 16 bits of) the end of the current block is the start of the next
// block -- what htslib's bgzf_tell reports -- so that it compares correctly with chunk ends of the index.
uint64_t bg_tell(const tredbam* b) {
    if (b->block_clen > 0 && b->upos >= b->block_size) return (uint64_t)(b->block_coffset + b->block_clen) << 16;
    return ((uint64_t)b->block_coffset << 16) | (uint64_t)b->upos;
}

// read n bytes; returns the number actually read (short at end of file), <0 on error
int64_t bg_read(tredbam* b, uint8_t* dst, int64_t n) {
    int64_t done = 0;
    while (n > 0) {
        if (b->upos >= b->block_size) {
            const int rc = load_block(b, b->block_coffset + b->block_clen);
            if (rc < 0) return rc;
            b->upos = 0;
            if (rc == 0) break;            // end of file
            if (b->block_size == 0) continue;  // empty block (e.g. the EOF marker): move on
        }
        const int64_t take = std::min<int64_t>(n, (int64_t)(b->block_size - b->upos));
        memcpy(dst + done, b->block + b->upos, (size_t)take);
        b->upos += (size_t)take;
        done += take;
        n -= take;
    }
    return done;
}

// next alignment record (without its 4-byte block_size) at b->recp, b->rec_size bytes: in place when it lies inside
// the current block, assembled in b->rec when it straddles blocks; 1 ok, 0 end of file, <0 error
int next_record(tredbam* b) {
    if (b->upos + 4 <= b->block_size) {
        const int32_t size = (int32_t)le32(b->block + b->upos);
        if (size < 32) return fail(b, -8, "bad alignment record size %d", size);
        if (b->upos + 4 + (size_t)size <= b->block_size) {
            b->recp = b->block + b->upos + 4;
            b->rec_size = (size_t)size;
            b->upos += 4 + (size_t)size;
            return 1;
        }
    }
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
    b->recp = b->rec.data();
    b->rec_size = (size_t)size;
    return 1;
}

const bool CIGAR_REF[16] = {true, false, true, true, false, false, false, true, true};   // MIDNSHP=X

// reference_end of the current record (-1: none) without building the output record: what the walks that only look
// at positions need (depth, pair lengths, mate fields -- 3 of 4 records visited by a scan)
inline int record_end(tredbam* b, int32_t* endp) {
    const uint8_t* r = b->recp;
    const size_t l_name = r[8], n_cigar = le16(r + 12);
    const int32_t l_seq = (int32_t)le32(r + 16);
    if (l_seq < 0 || 32 + l_name + 4 * n_cigar + ((size_t)l_seq + 1) / 2 > b->rec_size)
        return fail(b, -8, "alignment record shorter than its fields");
    int32_t end = -1;
    if (!(le16(r + 14) & 0x4) && n_cigar > 0) {
        const uint8_t* cig = r + 32 + l_name;
        int64_t e = (int32_t)le32(r + 4);
        for (size_t k = 0; k < n_cigar; ++k) {
            const uint32_t c = le32(cig + 4 * k);
            if (CIGAR_REF[c & 15]) e += c >> 4;
        }
        end = (int32_t)e;
    }
    *endp = end;
    return 0;
}

// append b->rec to b->out in the tredbam_rec layout; returns reference_end (-1: none) through *endp
int emit_record(tredbam* b, int32_t* endp, bool store) {
    const uint8_t* r = b->recp;
    const size_t size = b->rec_size;
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
    if (b->bai_mapped) { munmap(const_cast<uint8_t*>(b->bai_p), b->bai_n); b->bai_mapped = false; }   // (an earlier, refused index)
    int fd = -1;
    for (const std::string& c : cand)
        if ((fd = open(c.c_str(), O_RDONLY | O_CLOEXEC)) >= 0) break;
    if (fd < 0) return fail(b, -4, "no .bai index next to %s", b->path.c_str());
    struct Bytes { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } } d{nullptr, 0};
    {
        struct stat st;
        void* m = MAP_FAILED;
        if (fstat(fd, &st) == 0 && st.st_size > 0) m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            b->bai_p = (const uint8_t*)m;
            b->bai_n = (size_t)st.st_size;
            b->bai_mapped = true;
        } else {                                           // (no mapping: the bytes are read)
            std::vector<uint8_t>& v = b->bai;
            v.clear();
            uint8_t tmp[65536];
            ssize_t g;
            while ((g = read(fd, tmp, sizeof tmp)) > 0) v.insert(v.end(), tmp, tmp + g);
            b->bai_p = v.data();
            b->bai_n = v.size();
        }
        d = Bytes{b->bai_p, b->bai_n};
    }
    close(fd);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return fail(b, -4, "bad BAI magic");
    const int32_t n_ref = (int32_t)le32(d.data() + 4);
    size_t p = 8;
    b->index.assign((size_t)std::max(n_ref, 0), RefIndex());
    for (int32_t t = 0; t < n_ref; ++t) {
        if (p + 4 > d.size()) return fail(b, -4, "truncated BAI");
        const int32_t n_bin = (int32_t)le32(d.data() + p); p += 4;
        if (n_bin < 0) return fail(b, -4, "truncated BAI");
        b->index[t].bins_at = p;
        b->index[t].n_bin = n_bin;
        for (int32_t k = 0; k < n_bin; ++k) {          // (only the chunk counts are looked at here)
            if (p + 8 > d.size()) return fail(b, -4, "truncated BAI");
            const int32_t n_chunk = (int32_t)le32(d.data() + p + 4); p += 8;
            if (n_chunk < 0 || p + 16 * (size_t)n_chunk > d.size()) return fail(b, -4, "truncated BAI");
            p += 16 * (size_t)n_chunk;
        }
        if (p + 4 > d.size()) return fail(b, -4, "truncated BAI");
        const int32_t n_intv = (int32_t)le32(d.data() + p); p += 4;
        if (n_intv < 0 || p + 8 * (size_t)n_intv > d.size()) return fail(b, -4, "truncated BAI");
        b->index[t].lin = d.data() + p;
        b->index[t].n_lin = (size_t)n_intv;
        p += 8 * (size_t)n_intv;
    }
    b->index_loaded = true;
    return 0;
}

// the bins of one contig, parsed on first use (load_index checked the lengths)
const RefIndex& contig_index(tredbam* b, int32_t tid) {
    RefIndex& ix = b->index[(size_t)tid];
    if (!ix.bins_parsed) {
        const uint8_t* d = b->bai_p;
        size_t p = ix.bins_at;
        ix.bins.reserve((size_t)ix.n_bin * 2);
        for (int32_t k = 0; k < ix.n_bin; ++k) {
            const uint32_t bin = le32(d + p);
            const int32_t n_chunk = (int32_t)le32(d + p + 4); p += 8;
            auto& v = ix.bins[bin];
            v.reserve((size_t)n_chunk);
            for (int32_t c = 0; c < n_chunk; ++c, p += 16) v.emplace_back(le64(d + p), le64(d + p + 8));
        }
        ix.bins_parsed = true;
    }
    return ix;
}

// The merged index chunks (virtual offsets) a query of [start, end) on tid reads, like htslib's: reg2bins of the 5-level
// scheme, every chunk cut at the linear index' offset of the 16 kb window that holds `start`.  0, or <0.
int region_chunks(tredbam* b, int32_t tid, int64_t& start, int64_t& end, std::vector<std::pair<uint64_t, uint64_t>>& merged) {
    if (tid < 0 || tid >= (int32_t)b->ref_names.size()) return fail(b, -2, "invalid contig id %d", tid);
    start = std::max<int64_t>(0, start);
    if (end < 0) end = b->ref_lens[tid];
    if (start > end) return fail(b, -2, "invalid coordinates: start > end");
    int rc = load_index(b);
    if (rc) return rc;
    merged.clear();
    if (tid >= (int32_t)b->index.size()) return 0;
    const RefIndex& ix = contig_index(b, tid);
    uint64_t min_off = 0;
    if (ix.n_lin > 0) min_off = ix.linear(std::min<size_t>((size_t)(start >> 14), ix.n_lin - 1));
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
    for (const auto& ch : chunks) {
        if (!merged.empty() && ch.first <= merged.back().second) merged.back().second = std::max(merged.back().second, ch.second);
        else merged.push_back(ch);
    }
    return 0;
}

// Region walk shared by fetch and the depth sum.  visit(end) is called for every record overlapping the region
// after emit_record() has parsed it (and stored it when `store`).
template <typename F>
int64_t walk_chunks(tredbam* b, int32_t tid, int64_t start, int64_t end, const std::vector<std::pair<uint64_t, uint64_t>>& merged,
                    bool store, F visit) {
    int rc;
    int64_t n = 0;
    for (const auto& ch : merged) {
        if ((rc = bg_seek(b, ch.first)) < 0) return rc;
        while (bg_tell(b) < ch.second) {
            rc = next_record(b);
            if (rc < 0) return rc;
            if (rc == 0) break;
            const int32_t rtid = (int32_t)le32(b->recp);
            const int32_t rpos = (int32_t)le32(b->recp + 4);
            if (rtid != tid || rpos >= end) {
                if (rtid > tid || (rtid == tid && rpos >= end)) break;
                continue;
            }
            // overlap test first (cheap pass without storing), then store
            const size_t mark = b->out.size();
            int32_t rend = -1;
            if ((rc = store ? emit_record(b, &rend, true) : record_end(b, &rend)) < 0) return rc;
            int64_t e = rend;
            if (rend < 0 || rend <= rpos) e = (int64_t)rpos + 1;   // placed-unmapped / zero length: one base (bam_endpos)
            if (e > start && visit(rend, rpos, le16(b->recp + 14), b->recp)) ++n;
            else b->out.resize(mark);
        }
    }
    return n;
}

template <typename F>
int64_t walk_region(tredbam* b, int32_t tid, int64_t start, int64_t end, bool store, F visit) {
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    const int rc = region_chunks(b, tid, start, end, merged);
    if (rc) return rc;
    return walk_chunks(b, tid, start, end, merged, store, visit);
}

// The records of [start, end) that lie between the virtual offsets vbeg and vend of the walk over the LARGER region
// [outer_start, outer_end) on the same contig: a walker that went over the larger region before (the device's pair walk)
// says where the first and behind the last record of the smaller one are, and this walk reads only the blocks in
// between.  Every record of the small region is a record of the large one's chunks, in the same order.
template <typename F>
int64_t walk_region_between(tredbam* b, int32_t tid, int64_t outer_start, int64_t outer_end, int64_t start, int64_t end,
                            uint64_t vbeg, uint64_t vend, F visit) {
    std::vector<std::pair<uint64_t, uint64_t>> merged, cut;
    const int rc = region_chunks(b, tid, outer_start, outer_end, merged);
    if (rc) return rc;
    start = std::max<int64_t>(0, start);
    for (const auto& ch : merged) {
        const uint64_t lo = std::max(ch.first, vbeg), hi = std::min(ch.second, vend);
        if (lo < hi) cut.emplace_back(lo, hi);
    }
    return walk_chunks(b, tid, start, end, cut, false, visit);
}


// The paired-end selection behind tredbam_pe_lengths / tredbam_scan (see include/tredbam.h for the rules): records are
// fed one by one (add), the lengths come out at the end (finish).
struct PairTable {
    struct Mate { int32_t pos, end, lead_clip, trail_clip; bool reverse; };
    struct Pair { int n; uint32_t name_at, name_len; Mate m[2]; };
    // query name -> pair, in order of first appearance (the reference walks a dict in that order): an open-addressing
    // table over a 64-bit hash of the name (hash_name), names kept in one pool for the equality check (a +-10 kb window
    // holds ~4 000 records; a std::string and a node allocation per record were a third of the scan's parse time)
    std::vector<Pair> pairs;
    std::vector<char> pool;
    std::vector<int32_t> table = std::vector<int32_t>(1 << 13, -1);
    size_t mask = (1 << 13) - 1;

    // eight name bytes per multiply (a byte-wise FNV-1a over ~20-byte names was a third of the walk's time: 130 000
    // records per sample, one dependent multiply per byte)
    static uint64_t hash_name(const char* p, uint32_t len) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
        for (; len >= 8; p += 8, len -= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 32;
        }
        if (len) {
            uint64_t w = 0;
            memcpy(&w, p, len);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 32;
        }
        return h;
    }

    Pair& find_or_add(const char* name, uint32_t len) {
        const uint64_t h = hash_name(name, len);
        if ((pairs.size() + 1) * 2 > table.size()) {          // keep the load below one half
            table.assign(table.size() * 2, -1);
            mask = table.size() - 1;
            for (size_t k = 0; k < pairs.size(); ++k) {
                const uint64_t g = hash_name(pool.data() + pairs[k].name_at, pairs[k].name_len);
                size_t at = (size_t)(g ^ (g >> 29)) & mask;
                while (table[at] >= 0) at = (at + 1) & mask;
                table[at] = (int32_t)k;
            }
        }
        size_t at = (size_t)(h ^ (h >> 29)) & mask;
        for (;; at = (at + 1) & mask) {
            const int32_t k = table[at];
            if (k < 0) break;
            const Pair& q = pairs[(size_t)k];
            if (q.name_len == len && memcmp(pool.data() + q.name_at, name, len) == 0) return pairs[(size_t)k];
        }
        table[at] = (int32_t)pairs.size();
        Pair p{};
        p.name_at = (uint32_t)pool.size();
        p.name_len = len;
        pool.insert(pool.end(), name, name + len);
        pairs.push_back(p);
        return pairs.back();
    }

    void add(int32_t rend, int32_t rpos, uint16_t flag, const uint8_t* r) {
        if (!(flag & 0x1) || (flag & 0x4) || (flag & 0x400)) return;   // paired, mapped, not a duplicate
        const int l_name = r[8];
        const int n_cigar = le16(r + 12);
        Pair& p = find_or_add((const char*)r + 32, (uint32_t)std::max(l_name - 1, 0));
        if (p.n < 2) {
            Mate& m = p.m[p.n];
            m.pos = rpos;
            m.end = rend;
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
    }

    int finish(tredbam* b, int64_t tstart, int64_t tend, int32_t span, std::vector<int32_t>& global_lens,
               std::vector<int32_t>& target_lens) const {
        for (const Pair& p : pairs) {
            if (p.n < 2) continue;
            const Mate &a = p.m[0], &bb = p.m[1];
            if (a.reverse || !bb.reverse) continue;            // mapped in +, - orientation
            if (bb.end < 0) return fail(b, -9, "paired read without an alignment end in the window");
            const int64_t tlen = ((int64_t)bb.end + bb.trail_clip) - ((int64_t)a.pos - a.lead_clip);
            if (tlen >= span) continue;
            if (a.pos < tstart && bb.end > tend) target_lens.push_back((int32_t)tlen);
            else global_lens.push_back((int32_t)tlen);
        }
        return 0;
    }
};

int pair_lengths(tredbam* b, int32_t tid, int64_t start, int64_t end, int64_t tstart, int64_t tend, int32_t span,
                 std::vector<int32_t>& global_lens, std::vector<int32_t>& target_lens) {
    PairTable pt;
    const int64_t n = walk_region(b, tid, start, end, false, [&](int32_t rend, int32_t rpos, uint16_t flag, const uint8_t* r) {
        pt.add(rend, rpos, flag, r);
        return true;
    });
    if (n < 0) return (int)n;
    return pt.finish(b, tstart, tend, span, global_lens, target_lens);
}

// One selected read into the scan pools: 2-bit codes + N mask in libtredgpu's read layout (tredgpu.h: ceil(L/16)
// words of codes, base k in bits 2k..2k+1, then ceil(L/32) words of N flags), the raw 4-bit sequence for the
// JSON `details`, the query name, and the index of that name among the unit's names (the pair id of
// --norepeatpairs).
void pool_read(tredbam* b, const uint8_t* r, std::unordered_map<std::string, int32_t>& names) {
    const int l_name = r[8];
    const int n_cigar = le16(r + 12);
    const int32_t L = (int32_t)le32(r + 16);
    const uint8_t* seq = r + 32 + l_name + 4 * (size_t)n_cigar;
    const size_t nb = ((size_t)L + 15) >> 4, nm = ((size_t)L + 31) >> 5;
    const size_t w0 = b->sc_packed.size();
    b->sc_packed.resize(w0 + nb + nm, 0u);
    uint32_t* rec = b->sc_packed.data() + w0;
    static const int8_t CODE[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   // "=ACMGRSVTWYHKDBN"
    for (int32_t i = 0; i < L; ++i) {
        const uint8_t v = seq[i >> 1];
        const int code = CODE[(i & 1) ? (v & 15) : (v >> 4)];
        if (code == 4) rec[nb + ((size_t)i >> 5)] |= 1u << (i & 31);
        else rec[(size_t)i >> 4] |= (uint32_t)code << ((i & 15) * 2);
    }
    if (L == 0) b->sc_noseq = true;          // (SEQ '*': pysam's query_sequence is None and the reference's len(seq) raises)
    b->sc_word_off.push_back((int64_t)b->sc_packed.size());
    b->sc_read_len.push_back(L);
    b->sc_seq4.insert(b->sc_seq4.end(), seq, seq + ((size_t)L + 1) / 2);
    b->sc_seq4_off.push_back((int64_t)b->sc_seq4.size());
    const char* nm_p = (const char*)r + 32;
    const size_t nlen = (size_t)std::max(l_name - 1, 0);
    b->sc_names.insert(b->sc_names.end(), nm_p, nm_p + nlen);
    b->sc_name_off.push_back((int64_t)b->sc_names.size());
    const auto it = names.emplace(std::string(nm_p, nlen), (int32_t)names.size()).first;
    b->sc_name_id.push_back(it->second);
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
    {
        struct stat st;
        if (fstat(fileno(fp), &st) == 0 && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (m != MAP_FAILED) { b->map = (const uint8_t*)m; b->map_size = (size_t)st.st_size; }
        }
    }
    auto bail = [&](int code) {
        g_open_error = b->err;
        if (b->map) munmap(const_cast<uint8_t*>(b->map), b->map_size);
        fclose(b->fp);
        delete b;
        return code;
    };
    int rc = bg_seek(b, 0);
    if (rc < 0) return bail(rc);
    b->err.clear();
    auto truncated = [&]() {                       // keeps the block layer's reason (CRC mismatch, bad block, ...)
        b->err = b->err.empty() ? std::string("truncated BAM header") : "truncated BAM header: " + b->err;
        return bail(-6);
    };
    uint8_t w[8];
    if (bg_read(b, w, 4) != 4 || memcmp(w, "BAM\1", 4) != 0) { b->err = "not a BAM file: " + p; return bail(-6); }
    if (bg_read(b, w, 4) != 4) return truncated();
    const int32_t l_text = (int32_t)le32(w);
    std::vector<uint8_t> text((size_t)std::max(l_text, 0));
    if (bg_read(b, text.data(), l_text) != l_text || bg_read(b, w, 4) != 4) return truncated();
    const int32_t n_ref = (int32_t)le32(w);
    for (int32_t t = 0; t < n_ref; ++t) {
        if (bg_read(b, w, 4) != 4) return truncated();
        const int32_t l_name = (int32_t)le32(w);
        std::vector<uint8_t> nm((size_t)std::max(l_name, 1));
        if (bg_read(b, nm.data(), l_name) != l_name || bg_read(b, w, 4) != 4) return truncated();
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
    if (b->map) munmap(const_cast<uint8_t*>(b->map), b->map_size);
    if (b->bai_mapped) munmap(const_cast<uint8_t*>(b->bai_p), b->bai_n);
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
    std::vector<int32_t> g, t;
    const int rc = pair_lengths(b, tid, start, end, tstart, tend, span, g, t);
    if (rc < 0) return rc;
    for (size_t i = 0; i < g.size() && global_lens && (int64_t)i < cap_global; ++i) global_lens[i] = g[i];
    for (size_t i = 0; i < t.size() && target_lens && (int64_t)i < cap_target; ++i) target_lens[i] = t[i];
    *n_global = (int64_t)g.size();
    *n_target = (int64_t)t.size();
    return 0;
}

int tredbam_inflate_raw(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t out_len) {
    if (!in || !out || n_in < 0 || out_len < 0) return -2;
    static thread_local tredbam_inflate::Tables tables;
    std::vector<uint8_t> buf((size_t)out_len + tredbam_inflate::SLACK);
    if (!inflate_dispatch(in, (size_t)n_in, buf.data(), (size_t)out_len, tables)) return 0;
    memcpy(out, buf.data(), (size_t)out_len);
    return 1;
}

uint32_t tredbam_crc32(uint32_t crc, const uint8_t* buf, int64_t len) {
    return (buf && len > 0) ? tredbam_crc::crc32(crc, buf, (size_t)len) : crc;
}

int64_t tredbam_details_json(const uint8_t* seq4, const int64_t* seq4_off, const int32_t* read_len, const char* names,
                             const int64_t* name_off, const int64_t* reads, const uint8_t* tags, const int32_t* hs,
                             int64_t n, char* out, int64_t cap) {
    if (n < 0 || !out || cap < 2 || (n > 0 && (!seq4 || !seq4_off || !read_len || !names || !name_off || !reads || !tags || !hs)))
        return -2;
    static const char* const TAGS[6] = {nullptr, "FULL", "PREF", "POST", "REPT", "HANG"};
    static const char BASES[] = "=ACMGRSVTWYHKDBN";
    char* p = out;
    char* const end = out + cap;
    auto put = [&](const char* s, size_t len) { memcpy(p, s, len); p += len; };
#define TREDBAM_LIT(S) put(S, sizeof(S) - 1)
    if (n == 0) { TREDBAM_LIT("[]"); return p - out; }
    TREDBAM_LIT("[\n");
    for (int64_t i = 0; i < n; ++i) {
        const int64_t rd = reads[i];
        const int64_t n0 = name_off[rd], n1 = name_off[rd + 1];
        const int32_t L = read_len[rd];
        if (tags[i] < 1 || tags[i] > 5 || L < 0 || n1 < n0) return -2;
        // worst case of this element: fixed text < 160, the name doubled by escapes, the bases, 11 digits
        if ((int64_t)(end - p) < 200 + 2 * (n1 - n0) + L) return -3;
        TREDBAM_LIT("            {\n                \"h\": ");
        p += snprintf(p, 16, "%d", (int)hs[i]);
        TREDBAM_LIT(",\n                \"id\": \"");
        for (int64_t k = n0; k < n1; ++k) {
            const unsigned char c = (unsigned char)names[k];
            if (c < 0x20 || c >= 0x7f) return -1;
            if (c == '"' || c == '\\') *p++ = '\\';
            *p++ = (char)c;
        }
        TREDBAM_LIT("\",\n                \"seq\": \"");
        const uint8_t* sq = seq4 + seq4_off[rd];
        {   // two bases per byte of the 4-bit sequence: one table look-up and one 2-byte store per pair
            static const struct Pairs { char t[256][2]; Pairs() { for (int b = 0; b < 256; ++b) { t[b][0] = BASES[b >> 4]; t[b][1] = BASES[b & 15]; } } } PAIRS;
            const int32_t whole = L >> 1;
            for (int32_t k = 0; k < whole; ++k, p += 2) memcpy(p, PAIRS.t[sq[k]], 2);
            if (L & 1) *p++ = BASES[sq[whole] >> 4];
        }
        TREDBAM_LIT("\",\n                \"tag\": \"");
        put(TAGS[tags[i]], 4);
        TREDBAM_LIT("\"\n            }");
        if (i + 1 < n) TREDBAM_LIT(",\n");
    }
    if (end - p < 16) return -3;
    TREDBAM_LIT("\n        ]");
#undef TREDBAM_LIT
    return p - out;
}

// Python's repr(float) (float_repr_style 'short'): the shortest digit string that round-trips -- std::to_chars
// produces the same digits -- laid out in exponent form when the decimal exponent is <= -5 or >= 16, else positionally
// with ".0" appended to integers.
int tredbam_float_repr(double x, char* out) {
    if (!out || !std::isfinite(x)) return -1;
    char* p = out;
    if (std::signbit(x)) { *p++ = '-'; x = -x; }
    if (x == 0) { memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
    char sci[40];
    const auto res = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
    // d[.ddd]e[+-]XX[X]
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < res.ptr && *q != 'e'; ++q) if (*q != '.') digits[nd++] = *q;
    int e10 = 0;
    {
        const char* r = q + 1;
        const bool neg = *r == '-';
        ++r;
        for (; r < res.ptr; ++r) e10 = e10 * 10 + (*r - '0');
        if (neg) e10 = -e10;
    }
    if (e10 <= -5 || e10 >= 16) {                      // exponent form: to_chars' own text is Python's
        const size_t len = (size_t)(res.ptr - sci);
        memcpy(p, sci, len);
        return (int)(p - out) + (int)len;
    }
    if (e10 >= 0) {
        for (int i = 0; i <= e10; ++i) *p++ = i < nd ? digits[i] : '0';
        *p++ = '.';
        if (nd > e10 + 1) for (int i = e10 + 1; i < nd; ++i) *p++ = digits[i];
        else *p++ = '0';
    } else {
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -e10 - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = digits[i];
    }
    return (int)(p - out);
}

namespace {
inline int key_int(char* out, int32_t v) {                 // decimal text of v (at most 11 bytes), its length
    char buf[12];
    char* p = buf + sizeof buf;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    const int n = (int)(buf + sizeof buf - p);
    memcpy(out, p, (size_t)n);
    return n;
}
}  // namespace

int64_t tredbam_sparse_json(const int32_t* a, const int32_t* b, const double* values, int64_t n, int32_t depth,
                            char* out, int64_t cap) {
    if (n < 0 || depth < 0 || depth > 16 || !out || cap < 2 || (n > 0 && (!a || !values))) return -2;
    char* p = out;
    char* const end = out + cap;
    if (n == 0) { memcpy(p, "{}", 2); return 2; }
    struct Item { char key[24]; int klen; int64_t at; };
    std::vector<Item> items((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        Item& it = items[(size_t)i];
        it.klen = key_int(it.key, a[i]);
        if (b) { it.key[it.klen++] = ','; it.klen += key_int(it.key + it.klen, b[i]); }
        it.at = i;
    }
    std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) {
        const int c = memcmp(x.key, y.key, (size_t)std::min(x.klen, y.klen));
        return c != 0 ? c < 0 : x.klen < y.klen;
    });
    for (size_t i = 1; i < items.size(); ++i)
        if (items[i].klen == items[i - 1].klen && memcmp(items[i].key, items[i - 1].key, (size_t)items[i].klen) == 0) return -1;
    const int pad = 4 * (depth + 1);
    *p++ = '{'; *p++ = '\n';
    for (size_t i = 0; i < items.size(); ++i) {
        if (end - p < pad + 80) return -3;
        memset(p, ' ', (size_t)pad); p += pad;
        *p++ = '"';
        memcpy(p, items[i].key, (size_t)items[i].klen); p += items[i].klen;
        *p++ = '"'; *p++ = ':'; *p++ = ' ';
        const int len = tredbam_float_repr(values[items[i].at], p);
        if (len < 0) return -1;
        p += len;
        if (i + 1 < items.size()) *p++ = ',';
        *p++ = '\n';
    }
    if (end - p < 4 * depth + 2) return -3;
    memset(p, ' ', (size_t)(4 * depth)); p += 4 * depth;
    *p++ = '}';
    return p - out;
}

// Many distributions / many details lists in one call (a sample's 90 + 30 of them: the per-call cost of the binding
// was a third of the driver thread's time per sample).  Item k covers entries off[k] .. off[k+1] of the input arrays;
// its text lands at out[out_off[k] .. out_off[k+1]); status[k] = 0, or the single call's -1 (text left empty: the
// caller's generic encoder takes that item).  Returns the bytes written, -3 when cap is too small, -2 on bad arguments.
int64_t tredbam_sparse_json_many(const int32_t* a, const int32_t* b, const double* values, const int64_t* off,
                                 const uint8_t* two_part, int64_t n_items, int32_t depth, char* out, int64_t cap,
                                 int64_t* out_off, int8_t* status) {
    if (n_items < 0 || !off || !two_part || !out || !out_off || !status) return -2;
    int64_t at = 0;
    out_off[0] = 0;
    for (int64_t k = 0; k < n_items; ++k) {
        const int64_t n = off[k + 1] - off[k];
        if (n < 0 || (two_part[k] && !b)) return -2;
        const int64_t got = tredbam_sparse_json(a ? a + off[k] : nullptr, two_part[k] ? b + off[k] : nullptr,
                                                values ? values + off[k] : nullptr, n, depth, out + at, cap - at);
        if (got == -3 || got == -2) return got;
        status[k] = got < 0 ? -1 : 0;
        if (got > 0) at += got;
        out_off[k + 1] = at;
    }
    return at;
}

int64_t tredbam_details_json_many(const uint8_t* seq4, const int64_t* seq4_off, const int32_t* read_len, const char* names,
                                  const int64_t* name_off, const int64_t* reads, const uint8_t* tags, const int32_t* hs,
                                  const int64_t* off, int64_t n_items, char* out, int64_t cap, int64_t* out_off,
                                  int8_t* status) {
    if (n_items < 0 || !off || !out || !out_off || !status) return -2;
    int64_t at = 0;
    out_off[0] = 0;
    for (int64_t k = 0; k < n_items; ++k) {
        const int64_t n = off[k + 1] - off[k];
        if (n < 0) return -2;
        const int64_t got = tredbam_details_json(seq4, seq4_off, read_len, names, name_off, reads ? reads + off[k] : nullptr,
                                                 tags ? tags + off[k] : nullptr, hs ? hs + off[k] : nullptr, n, out + at,
                                                 cap - at);
        if (got == -3 || got == -2) return got;
        status[k] = got < 0 ? -1 : 0;
        if (got > 0) at += got;
        out_off[k + 1] = at;
    }
    return at;
}

// Mean, population standard deviation and the 40-bin histogram over [0, 1000] of many slices of a pair-length pool
// (the numbers behind the JSON's PEG / PET / P_PEG / P_PET strings, models.py:87-98 of the reference: mean_std and
// histogram per list), all loci of a sample in one pass.  Sums run in pool order in double precision.
int tredbam_pair_stats(const int32_t* pool, const int64_t* first, const int32_t* count, int64_t n, double* mean,
                       double* sd, int32_t* hist) {
    if (n < 0 || (n > 0 && (!first || !count || !mean || !sd || !hist))) return -2;
    constexpr int BINS = 40, SPAN = 1000, WIDTH = SPAN / BINS;
    for (int64_t k = 0; k < n; ++k) {
        const int32_t c = count[k];
        int32_t* h = hist + k * BINS;
        for (int j = 0; j < BINS; ++j) h[j] = 0;
        mean[k] = sd[k] = 0;
        if (c <= 0) continue;
        if (!pool) return -2;
        const int32_t* x = pool + first[k];
        double sum = 0;
        for (int32_t i = 0; i < c; ++i) sum += (double)x[i];
        const double m = sum / (double)c;
        double q = 0;
        for (int32_t i = 0; i < c; ++i) {
            const double d = (double)x[i] - m;
            q += d * d;
            if (x[i] >= 0 && x[i] <= SPAN) ++h[std::min(x[i] / WIDTH, BINS - 1)];
        }
        mean[k] = m;
        sd[k] = std::sqrt(q / (double)c);
    }
    return 0;
}

int tredbam_max_read_len(tredbam* b, int64_t first_n, int32_t* out) {
    if (!b || !out) return -2;
    int rc = bg_seek(b, b->first_record);
    if (rc < 0) return rc;
    int32_t best = -1;
    for (int64_t k = 0; first_n <= 0 || k < first_n; ++k) {
        rc = next_record(b);
        if (rc < 0) return rc;
        if (rc == 0) break;
        best = std::max(best, (int32_t)le32(b->recp + 16));
    }
    if (best < 0) return fail(b, -8, "no alignment records in %s", b->path.c_str());
    *out = best;
    return 0;
}

}  // extern "C"

namespace {
// The records of an alternative locus whose mate lies in the window, handed in as virtual offsets by a walker that went
// over the region elsewhere (tredbam_alt_result): every one is read where it is said to be and checked -- it is a record
// of this region with its mate in the window -- before any is pooled.  false: not what was promised (the region is then
// walked here as usual).
bool pool_walked(tredbam* b, const tredbam_region& a, int32_t mate_tid, int64_t win_lo, int64_t win_hi, const tredbam_alt_result& res,
                 std::unordered_map<std::string, int32_t>& names) {
    if (res.n < 0 || res.n > (int32_t)(sizeof res.vbeg / sizeof res.vbeg[0])) return false;
    const int64_t start = std::max<int64_t>(0, a.start);
    for (int pass = 0; pass < 2; ++pass)
        for (int32_t m = 0; m < res.n; ++m) {
            if (bg_seek(b, res.vbeg[m]) < 0 || next_record(b) <= 0) return false;
            const uint8_t* r = b->recp;
            if (pass == 0) {
                const int32_t rtid = (int32_t)le32(r), rpos = (int32_t)le32(r + 4);
                int32_t rend = -1;
                if (rtid != a.tid || rpos >= a.end || record_end(b, &rend) < 0) return false;
                const int64_t e = (rend < 0 || rend <= rpos) ? (int64_t)rpos + 1 : (int64_t)rend;
                const int32_t mt = (int32_t)le32(r + 20), mp = (int32_t)le32(r + 24);
                if (!(e > start) || mt != mate_tid || mp < win_lo || mp > win_hi) return false;
                if (m > 0 && res.vbeg[m] <= res.vbeg[m - 1]) return false;       // in file order, none twice
            } else
                pool_read(b, r, names);
        }
    return true;
}

// tredbam_scan / tredbam_scan_pe.  pe != nullptr: the pair lengths of site i are pe[i]'s slices of the two pools when
// pe[i].status == 0, and the walk over that site covers its window's records only (pe[i].win_vbeg .. win_vend); every other
// site is scanned as tredbam_scan scans it.
int scan_impl(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts, const tredbam_scan_opts* o,
              tredbam_unit* units, const tredbam_walk_result* pe, const int32_t* pe_global, const int32_t* pe_target,
              const tredbam_alt_result* alt_res) {
    if (!b || !o || n_sites < 0 || (n_sites > 0 && (!sites || !units))) return -2;
    b->out.clear();
    b->sc_packed.clear(); b->sc_read_len.clear(); b->sc_seq4.clear(); b->sc_names.clear(); b->sc_name_id.clear();
    b->sc_global.clear(); b->sc_target.clear();
    b->sc_word_off.assign(1, 0); b->sc_seq4_off.assign(1, 0); b->sc_name_off.assign(1, 0);
    for (int32_t i = 0; i < n_sites; ++i) {
        const tredbam_site& st = sites[i];
        tredbam_unit& u = units[i];
        memset(&u, 0, sizeof u);
        b->sc_noseq = false;
        u.read_first = (int64_t)b->sc_read_len.size();
        u.global_first = (int64_t)b->sc_global.size();
        u.target_first = (int64_t)b->sc_target.size();
        const int64_t win_lo = std::max<int64_t>(0, (int64_t)st.repeat_start - o->pad);
        const int64_t win_hi = (int64_t)st.repeat_end + o->pad;
        const int64_t pos_lo = std::max<int64_t>(0, (int64_t)st.repeat_start - o->readlen);
        const int64_t pos_hi = (int64_t)st.repeat_end + o->readlen;
        // ONE walk per locus: the +-pe_reach region of the pair lengths contains the window of the depth and of the
        // read selection, and a region query starts parsing at the beginning of its 16 kb index bin -- three separate
        // walks parsed ~11 000 records per locus to use ~4 500 of them.  A record of the large region is in the
        // window's query exactly when it starts before the window's end and ends behind its start (walk_region's own
        // test); depth, selection and pairing then see the records they saw before, in the same order.
        //   depth: pileup without truncation (see tredbam_pileup_depth_sum)
        //   selection: unmapped reads placed in the window (their mate is the anchor) and reads starting within one
        //   read length of the tract; then, from the alternative loci, reads whose MATE lies in the window
        std::unordered_map<std::string, int32_t> names;
        const bool with_pe = o->want_pe != 0;
        const int64_t p_lo = std::max<int64_t>((int64_t)st.repeat_start - o->pe_reach, 0);
        const int64_t p_hi = (int64_t)st.repeat_end + o->pe_reach;
        // pair lengths from a walker that went over the +-pe_reach region where the blocks were inflated: it also says
        // between which virtual offsets the window's records lie, and only those are walked here
        const bool hinted = pe && with_pe && pe[i].status == 0 && pe[i].n_global >= 0 && pe[i].n_target >= 0 &&
                            p_lo <= win_lo && p_hi >= win_hi;
        const bool wide = with_pe && !hinted && p_lo <= win_lo && p_hi >= win_hi;
        PairTable pt;
        int64_t depth_total = 0;
        auto each = [&](int32_t rend, int32_t rpos, uint16_t flag, const uint8_t* r) {
            int64_t e = rend;
            if (rend < 0 || rend <= rpos) e = (int64_t)rpos + 1;
            if (rpos < win_hi && e > win_lo) {                           // the window's query would return it
                if (o->want_depth && !(flag & (0x4 | 0x100 | 0x200 | 0x400)) && rend >= 0) depth_total += (int64_t)rend - rpos;
                if ((flag & 0x4) != 0 || (rpos >= pos_lo && rpos <= pos_hi)) pool_read(b, r, names);
            }
            if (wide) pt.add(rend, rpos, flag, r);
            return true;
        };
        const int64_t n = hinted ? walk_region_between(b, st.tid, p_lo, p_hi, win_lo, win_hi, pe[i].win_vbeg, pe[i].win_vend, each)
                                 : walk_region(b, st.tid, wide ? p_lo : win_lo, wide ? p_hi : win_hi, false, each);
        if (o->want_depth) {
            u.depth_status = n < 0 ? (int32_t)n : 0;
            u.depth_sum = n < 0 ? 0 : depth_total;
        }
        if (n == -2 || n == -4) u.status |= TREDBAM_UNIT_NO_FETCH;        // unknown contig / no index: no reads, go on
        else if (n < 0) u.status |= TREDBAM_UNIT_FAILED;
        if (n >= 0 && o->use_alts) {
            for (int32_t k = 0; k < st.n_alt; ++k) {
                const tredbam_region& a = alts[st.alt_first + k];
                if (a.tid < 0) continue;                                 // contig not in this file: skipped
                if (alt_res && alt_res[st.alt_first + k].status == 0 &&
                    pool_walked(b, a, st.tid, win_lo, win_hi, alt_res[st.alt_first + k], names))
                    continue;                                            // its records were found where the blocks were inflated
                walk_region(b, a.tid, a.start, a.end, false, [&](int32_t, int32_t, uint16_t, const uint8_t* r) {
                    const int32_t mate_tid = (int32_t)le32(r + 20), mate_pos = (int32_t)le32(r + 24);
                    if (mate_tid == st.tid && mate_pos >= win_lo && mate_pos <= win_hi) pool_read(b, r, names);
                    return true;
                });
            }
        }
        u.n_reads = (int32_t)((int64_t)b->sc_read_len.size() - u.read_first);
        if (b->sc_noseq) u.status |= TREDBAM_UNIT_NO_SEQ;
        // paired-end lengths around the tract
        if (with_pe && !(u.status & TREDBAM_UNIT_FAILED)) {
            const size_t g0 = b->sc_global.size(), t0 = b->sc_target.size();
            int rc;
            if (wide) rc = n < 0 ? (int)n : pt.finish(b, (int64_t)st.repeat_start - o->flank, (int64_t)st.repeat_end + o->flank,
                                                       o->span, b->sc_global, b->sc_target);
            else if (hinted) {
                rc = n < 0 ? (int)n : 0;                                    // computed elsewhere, in the same order
                if (rc == 0 && (pe[i].global_first < 0 || pe[i].target_first < 0 || pe[i].n_global < 0 || pe[i].n_target < 0 ||
                                (b->pe_pool_global >= 0 && pe[i].global_first + pe[i].n_global > b->pe_pool_global) ||
                                (b->pe_pool_target >= 0 && pe[i].target_first + pe[i].n_target > b->pe_pool_target)))
                    return fail(b, -2, "pair-length slices of site %d lie outside the pools handed in", (int)i);
                if (rc == 0) {
                    b->sc_global.insert(b->sc_global.end(), pe_global + pe[i].global_first, pe_global + pe[i].global_first + pe[i].n_global);
                    b->sc_target.insert(b->sc_target.end(), pe_target + pe[i].target_first, pe_target + pe[i].target_first + pe[i].n_target);
                }
            } else
                rc = pair_lengths(b, st.tid, p_lo, p_hi, (int64_t)st.repeat_start - o->flank,
                                  (int64_t)st.repeat_end + o->flank, o->span, b->sc_global, b->sc_target);
            if (rc < 0) { b->sc_global.resize(g0); b->sc_target.resize(t0); }
            u.pe_status = (rc == -2 || rc == -4) ? 0 : rc;     // unknown contig / no index: empty lists, as for the reads
            u.n_global = (int32_t)(b->sc_global.size() - g0);
            u.n_target = (int32_t)(b->sc_target.size() - t0);
        }
    }
    return 0;
}
}  // namespace

extern "C" {

int tredbam_scan(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                 const tredbam_scan_opts* o, tredbam_unit* units) {
    return scan_impl(b, sites, n_sites, alts, o, units, nullptr, nullptr, nullptr, nullptr);
}

int tredbam_pe_pool_sizes(tredbam* b, int64_t n_global, int64_t n_target) {
    if (!b) return -2;
    b->pe_pool_global = n_global;
    b->pe_pool_target = n_target;
    return 0;
}

int tredbam_scan_pe(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                    const tredbam_scan_opts* o, const tredbam_walk_result* pe, const int32_t* pe_global,
                    const int32_t* pe_target, tredbam_unit* units) {
    if (!pe || !pe_global || !pe_target) return -2;
    return scan_impl(b, sites, n_sites, alts, o, units, pe, pe_global, pe_target, nullptr);
}

int tredbam_scan_walked(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                        const tredbam_scan_opts* o, const tredbam_walk_result* pe, const int32_t* pe_global,
                        const int32_t* pe_target, const tredbam_alt_result* alt_res, tredbam_unit* units) {
    if (!pe || !pe_global || !pe_target || !alt_res) return -2;
    return scan_impl(b, sites, n_sites, alts, o, units, pe, pe_global, pe_target, alt_res);
}

// The walks over the alternative loci of tredbam_scan(sites, alts, o) as tasks for the same walker: task alt_first + k
// of site i is the region alts[alt_first + k]; a record counts when its mate lies on the site's contig (tstart) within
// [win_lo, win_hi] (the locus' window, both ends included: bam_parser.py:232-236).  n_chunks < 0: not walkable.
int64_t tredbam_plan_alt_walks(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts, int32_t n_alts,
                               const tredbam_scan_opts* o, tredbam_walk_task* tasks, tredbam_walk_chunk* chunks, int64_t cap_chunks) {
    if (!b || !o || n_sites < 0 || n_alts < 0 || (n_sites > 0 && !sites) || (n_alts > 0 && (!alts || !tasks)) || cap_chunks < 0 ||
        (cap_chunks > 0 && !chunks))
        return -2;
    std::unordered_map<int64_t, int32_t> index_of;
    index_of.reserve(b->plan.size() * 2);
    for (size_t k = 0; k < b->plan.size(); ++k) index_of[b->plan[k].coffset] = (int32_t)k;
    for (int32_t k = 0; k < n_alts; ++k) { memset(&tasks[k], 0, sizeof tasks[k]); tasks[k].n_chunks = -1; tasks[k].tid = -1; }
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    int64_t nc = 0;
    for (int32_t i = 0; i < n_sites; ++i) {
        const tredbam_site& st = sites[i];
        if (st.tid < 0 || !o->use_alts) continue;
        for (int32_t k = 0; k < st.n_alt; ++k) {
            if (st.alt_first + k < 0 || st.alt_first + k >= n_alts) return -2;
            const tredbam_region& a = alts[st.alt_first + k];
            tredbam_walk_task& t = tasks[st.alt_first + k];
            t.chunk_first = (int32_t)nc;
            t.block_end = (int32_t)b->plan.size();
            int64_t start = a.start, end = a.end;
            if (a.tid < 0 || region_chunks(b, a.tid, start, end, merged) != 0 || end > INT32_MAX) continue;
            if ((int64_t)merged.size() > cap_chunks - nc) return -3;
            t.tid = a.tid;
            t.start = (int32_t)start;
            t.end = (int32_t)end;
            t.tstart = st.tid;
            t.win_lo = (int32_t)std::max<int64_t>(0, (int64_t)st.repeat_start - o->pad);
            t.win_hi = st.repeat_end + o->pad;
            for (const auto& ch : merged) {
                tredbam_walk_chunk& c = chunks[nc++];
                const auto at = index_of.find((int64_t)(ch.first >> 16));
                c.begin_block = at == index_of.end() ? -1 : at->second;
                c.begin_upos = (int32_t)(ch.first & 0xFFFF);
                c.end_voffset = ch.second;
            }
            t.n_chunks = (int32_t)merged.size();
        }
    }
    return nc;
}

// The pair walks of tredbam_scan(sites, o) as tasks for a walker that holds the planned blocks inflated (the device:
// tredgpu_inflate_walk): per site the +-pe_reach region, its merged index chunks as (planned block, offset in it) ->
// end virtual offset.  Call after tredbam_plan with the same arguments.
int64_t tredbam_plan_walks(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_scan_opts* o,
                           tredbam_walk_task* tasks, tredbam_walk_chunk* chunks, int64_t cap_chunks) {
    if (!b || !o || n_sites < 0 || (n_sites > 0 && (!sites || !tasks)) || cap_chunks < 0 || (cap_chunks > 0 && !chunks)) return -2;
    std::unordered_map<int64_t, int32_t> index_of;
    index_of.reserve(b->plan.size() * 2);
    for (size_t k = 0; k < b->plan.size(); ++k) index_of[b->plan[k].coffset] = (int32_t)k;
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    int64_t nc = 0;
    for (int32_t i = 0; i < n_sites; ++i) {
        const tredbam_site& st = sites[i];
        tredbam_walk_task& t = tasks[i];
        memset(&t, 0, sizeof t);
        t.tid = st.tid;
        t.chunk_first = (int32_t)nc;
        t.n_chunks = -1;                                   // until proven walkable: the scan computes this site itself
        t.block_first = 0;
        t.block_end = (int32_t)b->plan.size();
        int64_t start = std::max<int64_t>((int64_t)st.repeat_start - o->pe_reach, 0), end = (int64_t)st.repeat_end + o->pe_reach;
        if (st.tid < 0 || region_chunks(b, st.tid, start, end, merged) != 0) continue;
        if (end > INT32_MAX || (int64_t)merged.size() > cap_chunks - nc) {
            if ((int64_t)merged.size() > cap_chunks - nc) return -3;
            continue;
        }
        t.start = (int32_t)start;
        t.end = (int32_t)end;
        t.tstart = st.repeat_start - o->flank;
        t.tend = st.repeat_end + o->flank;
        t.span = o->span;
        t.win_lo = (int32_t)std::max<int64_t>(0, (int64_t)st.repeat_start - o->pad);
        t.win_hi = st.repeat_end + o->pad;
        for (const auto& ch : merged) {
            tredbam_walk_chunk& c = chunks[nc++];
            const auto at = index_of.find((int64_t)(ch.first >> 16));
            c.begin_block = at == index_of.end() ? -1 : at->second;
            c.begin_upos = (int32_t)(ch.first & 0xFFFF);
            c.end_voffset = ch.second;
        }
        t.n_chunks = (int32_t)merged.size();
    }
    return nc;
}

// Plain regions as tasks for the same walker: task k = the records overlapping regions[k] = [start, end) of contig tid, its
// window the region itself, span 0 (the pair walk leaves it alone) -- the chrY windows of the sex inference (BamDepth.
// get_Y_depth, bam_parser.py:413-429), whose pile-up sums the device's read selection returns with the loci's.  Call after
// tredbam_plan with the regions among its `extra`.  n_chunks < 0: not walkable (the contig is not in the file, no index).
int64_t tredbam_plan_region_walks(tredbam* b, const tredbam_region* regions, int32_t n_regions, tredbam_walk_task* tasks,
                                  tredbam_walk_chunk* chunks, int64_t cap_chunks) {
    if (!b || n_regions < 0 || (n_regions > 0 && (!regions || !tasks)) || cap_chunks < 0 || (cap_chunks > 0 && !chunks)) return -2;
    std::unordered_map<int64_t, int32_t> index_of;
    index_of.reserve(b->plan.size() * 2);
    for (size_t k = 0; k < b->plan.size(); ++k) index_of[b->plan[k].coffset] = (int32_t)k;
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    int64_t nc = 0;
    for (int32_t i = 0; i < n_regions; ++i) {
        const tredbam_region& rg = regions[i];
        tredbam_walk_task& t = tasks[i];
        memset(&t, 0, sizeof t);
        t.tid = rg.tid;
        t.chunk_first = (int32_t)nc;
        t.n_chunks = -1;
        t.block_end = (int32_t)b->plan.size();
        int64_t start = rg.start, end = rg.end;
        if (rg.tid < 0 || region_chunks(b, rg.tid, start, end, merged) != 0 || end > INT32_MAX) continue;
        if ((int64_t)merged.size() > cap_chunks - nc) return -3;
        t.start = t.win_lo = (int32_t)start;
        t.end = t.win_hi = (int32_t)end;
        for (const auto& ch : merged) {
            tredbam_walk_chunk& c = chunks[nc++];
            const auto at = index_of.find((int64_t)(ch.first >> 16));
            c.begin_block = at == index_of.end() ? -1 : at->second;
            c.begin_upos = (int32_t)(ch.first & 0xFFFF);
            c.end_voffset = ch.second;
        }
        t.n_chunks = (int32_t)merged.size();
    }
    return nc;
}

// Per planned block, in the plan's (file) order: where it starts in the file, how long it is there, the CRC-32 its
// trailer promises, and whether the scan reads it in any case (alternative loci, extra regions) when the pair lengths
// and the windows' offsets come from elsewhere (tredbam_scan_pe).
int64_t tredbam_plan_blocks(tredbam* b, int64_t* coffset, int32_t* clen, uint32_t* crc, uint8_t* host) {
    if (!b) return -2;
    for (size_t k = 0; k < b->plan.size(); ++k) {
        const tredbam::Planned& p = b->plan[k];
        if (coffset) coffset[k] = p.coffset;
        if (clen) clen[k] = (int32_t)p.clen;
        if (crc) crc[k] = p.crc;
        if (host) host[k] = p.host;
    }
    return (int64_t)b->plan.size();
}

// ---- blocks inflated elsewhere (include/tredbam.h) ----------------------------------------------------------------
// The BGZF blocks the region walks of tredbam_scan(sites ...) will read: every block from the start of a merged chunk to
// its end -- or, sooner, to the linear index' offset two 16 kb windows behind the region's end when that window has a
// record of its own (records from there on start behind the region).  A block the plan leaves out is simply inflated by
// load_block when a walk gets there.
int64_t tredbam_plan(tredbam* b, const tredbam_site* sites, int32_t n_sites, const tredbam_region* alts,
                     const tredbam_scan_opts* o, const tredbam_region* extra, int32_t n_extra, int64_t* comp_bytes,
                     int64_t* out_bytes) {
    if (!b || !o || n_sites < 0 || (n_sites > 0 && !sites) || n_extra < 0 || (n_extra > 0 && !extra)) return -2;
    b->plan.clear();
    // `host`: a block the scan reads in any case when the pair lengths and the windows' offsets come from elsewhere
    // (tredbam_scan_pe): bit 0 those of the alternative loci (unless those walks are done elsewhere too:
    // tredbam_scan_walked), bit 1 those of the caller's extra regions.  Which blocks of a +-pe_reach region hold its
    // window's records is known only once the region has been walked (tredbam_walk_result)
    std::unordered_map<int64_t, size_t> seen;
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    auto add_region = [&](int32_t tid, int64_t start, int64_t end, uint8_t host) -> int {
        if (region_chunks(b, tid, start, end, merged) != 0) return 0;     // (the scan reports what is wrong with it)
        uint64_t cap = ~0ull;
        if (tid < (int32_t)b->index.size()) {
            const RefIndex& lin = b->index[tid];
            const size_t w = (size_t)((std::max<int64_t>(end, 1) - 1) >> 14) + 2;
            // (an entry equal to its predecessor is a window without a record of its own, filled in from before it)
            if (w < lin.n_lin && lin.linear(w) != 0 && lin.linear(w) > lin.linear(w - 1)) cap = lin.linear(w);
        }
        for (const auto& ch : merged) {
            const int64_t last = (int64_t)(std::min(ch.second, cap) >> 16);
            for (int64_t at = (int64_t)(ch.first >> 16); at <= last;) {
                const auto known = seen.find(at);
                if (known != seen.end()) {                                 // framed before: only the flag can change
                    tredbam::Planned& q = b->plan[known->second];
                    q.host = (uint8_t)(q.host | host);
                    at += q.clen;
                    continue;
                }
                BlockFrame fr = {};
                const uint8_t* comp = nullptr;
                const int rc = block_frame(b, at, fr, &comp);
                if (rc <= 0) break;                                        // end of file / damaged frame: left to the scan
                if (fr.isize > 0) {
                    seen[at] = b->plan.size();
                    b->plan.push_back({at, fr.payload_off, (int32_t)(fr.dlen - 8), fr.clen, fr.crc, fr.isize, host});
                }
                at += fr.clen;
            }
        }
        return 0;
    };
    for (int32_t i = 0; i < n_sites; ++i) {
        const tredbam_site& st = sites[i];
        const int64_t win_lo = std::max<int64_t>(0, (int64_t)st.repeat_start - o->pad), win_hi = (int64_t)st.repeat_end + o->pad;
        const int64_t p_lo = std::max<int64_t>((int64_t)st.repeat_start - o->pe_reach, 0), p_hi = (int64_t)st.repeat_end + o->pe_reach;
        if (st.tid < 0) continue;
        add_region(st.tid, win_lo, win_hi, 0);
        if (o->want_pe) add_region(st.tid, p_lo, p_hi, 0);
        if (o->use_alts && alts)
            for (int32_t k = 0; k < st.n_alt; ++k) {
                const tredbam_region& a = alts[st.alt_first + k];
                if (a.tid >= 0) add_region(a.tid, a.start, a.end, 1);
            }
    }
    for (int32_t k = 0; k < n_extra; ++k)          // other queries of the caller on this handle (the chrY depth windows)
        if (extra[k].tid >= 0) add_region(extra[k].tid, extra[k].start, extra[k].end, 2);
    // in file order: blocks that follow each other in the file then follow each other in the decoder's output, and a
    // record that straddles two of them lies there in one piece (the device's pair walk reads it in place)
    std::sort(b->plan.begin(), b->plan.end(), [](const tredbam::Planned& x, const tredbam::Planned& y) { return x.coffset < y.coffset; });
    int64_t cb = 0, ob = 0;
    for (const auto& p : b->plan) { cb += ((int64_t)p.payload_len + 3) & ~(int64_t)3; ob += p.isize; }
    if (comp_bytes) *comp_bytes = cb;
    if (out_bytes) *out_bytes = ob;
    return (int64_t)b->plan.size();
}

// The planned blocks' deflate payloads, copied to comp + comp_off[k] (4-byte aligned, from comp_base on) with the
// inflated sizes laid out from out_base on; comp_off / out_off receive the plan's n entries AND the end offsets as
// entry n (the next sample's bases).
int tredbam_plan_fill(tredbam* b, uint8_t* comp, int64_t comp_base, int64_t out_base, int64_t* comp_off, int64_t* out_off) {
    if (!b || !comp || !comp_off || !out_off || comp_base < 0 || (comp_base & 3) != 0 || out_base < 0) return -2;
    int64_t c = comp_base, o = out_base;
    for (size_t k = 0; k < b->plan.size(); ++k) {
        const tredbam::Planned& p = b->plan[k];
        const uint8_t* src = file_bytes(b, p.payload_off, (size_t)p.payload_len);
        if (!src) return fail(b, -6, "truncated BGZF block");
        comp_off[k] = c;
        out_off[k] = o;
        memcpy(comp + c, src, (size_t)p.payload_len);
        const int64_t padded = ((int64_t)p.payload_len + 3) & ~(int64_t)3;
        memset(comp + c + p.payload_len, 0, (size_t)(padded - p.payload_len));
        c += padded;
        o += p.isize;
    }
    comp_off[b->plan.size()] = c;
    out_off[b->plan.size()] = o;
    return 0;
}

// out + out_off[k] holds the planned block k inflated (status[k] == 0; others are left to load_block).  The memory
// stays the caller's and must live until tredbam_preload_clear or tredbam_close.
int tredbam_preload_crc(tredbam* b, const uint8_t* out, const int64_t* out_off, const int32_t* status, const uint32_t* crc) {
    if (!b || !out || !out_off || !status) return -2;
    b->preloaded.clear();
    b->preloaded.reserve(b->plan.size() * 2);
    b->preload_hits = b->preload_misses = 0;
    int32_t n = 0;
    for (size_t k = 0; k < b->plan.size(); ++k) {
        const tredbam::Planned& p = b->plan[k];
        if (status[k] != 0 || out_off[k + 1] - out_off[k] != (int64_t)p.isize) continue;
        if (crc && crc[k] != p.crc) continue;          // the decoder's checksum of its own output is not the trailer's
        // the decoder's CRC vouches for the bytes on the DEVICE; what the scan reads is their copy in host memory (a fetch
        // behind the walks).  One block in sixteen -- chosen by its place in the file -- is therefore checked again here at
        // first use (load_block): a stale or partial fetch does not pass unnoticed for long, at 1/16 of the CRC's cost.
        const bool sampled = (((uint64_t)p.coffset >> 4) * 0x9E3779B97F4A7C15ull >> 60) == 0;
        b->preloaded[p.coffset] = tredbam::Preloaded{out + out_off[k], p.isize, p.clen, p.crc, crc != nullptr && !sampled};
        ++n;
    }
    // blocks of an earlier scan in the handle's own cache stay valid; the current block pointer may not
    b->block_coffset = -1; b->block = nullptr; b->block_size = 0; b->block_clen = 0; b->upos = 0;
    return n;
}

int tredbam_preload(tredbam* b, const uint8_t* out, const int64_t* out_off, const int32_t* status) {
    return tredbam_preload_crc(b, out, out_off, status, nullptr);
}

void tredbam_preload_clear(tredbam* b, int64_t* hits, int64_t* misses) {
    if (!b) return;
    if (hits) *hits = b->preload_hits;
    if (misses) *misses = b->preload_misses;
    b->preloaded.clear();
    b->plan.clear();
    b->block_coffset = -1; b->block = nullptr; b->block_size = 0; b->block_clen = 0; b->upos = 0;
}

int tredbam_scan_pools(tredbam* b, tredbam_pools* p) {
    if (!b || !p) return -2;
    p->n_reads = (int64_t)b->sc_read_len.size();
    p->packed = b->sc_packed.data();
    p->n_words = (int64_t)b->sc_packed.size();
    p->word_off = b->sc_word_off.data();
    p->read_len = b->sc_read_len.data();
    p->seq4 = b->sc_seq4.data();
    p->seq4_off = b->sc_seq4_off.data();
    p->names = b->sc_names.data();
    p->name_off = b->sc_name_off.data();
    p->name_id = b->sc_name_id.data();
    p->global_lens = b->sc_global.data();
    p->n_global = (int64_t)b->sc_global.size();
    p->target_lens = b->sc_target.data();
    p->n_target = (int64_t)b->sc_target.size();
    return 0;
}

}  // extern "C"
