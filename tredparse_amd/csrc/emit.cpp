// libtredbam.so, second part: a sample's <key>.json and <key>.tred.vcf.gz written natively (include/tredbam.h,
// tredbam_emit_sample_files).
//
// What the reference does per sample in Python after its kernels (tredparse/tred.py:251-275 the tredCalls keys, :296-313
// to_json, :316-374 to_vcf; bam_parser.py:174-182, 248-287 tally / remove_pairs_of_rept; models.py:87-98 mean_std and
// histogram, :304-317 sparsify, :370-392 calc_label) and what tredparse_amd/tred.py's own Python path prints
// (format_scans + dumps_result + to_vcf) -- the same bytes, produced here from the batch's result arrays and the scan's
// pools without holding the interpreter lock.  Anything the fast printers do not cover (non-ASCII read names, duplicate
// distribution keys, values that are not finite, a sample whose BAM did not open) is handed back: the function returns 1
// and the Python path prints that sample.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "tredbam.h"

namespace {

thread_local std::string g_emit_error;

constexpr double SMALL_VALUE = 4.5399929762484854e-05;   // math.exp(-10) (models.py:34)
enum { TAG_NONE = 0, TAG_FULL = 1, TAG_PREF = 2, TAG_POST = 3, TAG_REPT = 4, TAG_HANG = 5 };

// numpy's float64 pairwise sum of a contiguous array (numpy/_core/src/umath/loops_utils.h.src DOUBLE_pairwise_sum):
// below 8 elements a plain loop; up to 128 eight running sums combined as ((0+1)+(2+3))+((4+5)+(6+7)); above, halves
// (the first a multiple of eight long).  `P.sum()` of sparsify (models.py:309) is this sum, added to an initial 0.0.
double pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

// json.dumps(str) with ensure_ascii: false on invalid UTF-8
bool json_string(std::string& out, const char* s) {
    static const char HEX[] = "0123456789abcdef";
    out += '"';
    const unsigned char* p = (const unsigned char*)s;
    auto esc = [&](uint32_t u) {
        out += "\\u";
        out += HEX[(u >> 12) & 15]; out += HEX[(u >> 8) & 15]; out += HEX[(u >> 4) & 15]; out += HEX[u & 15];
    };
    while (*p) {
        const unsigned char c = *p;
        if (c < 0x80) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                case '\b': out += "\\b"; break;
                case '\f': out += "\\f"; break;
                default:
                    if (c < 0x20) esc(c); else out += (char)c;
            }
            ++p;
            continue;
        }
        int extra;
        uint32_t cp;
        if ((c & 0xE0) == 0xC0) { extra = 1; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { extra = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { extra = 3; cp = c & 0x07; }
        else return false;
        for (int k = 1; k <= extra; ++k) {
            if ((p[k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[k] & 0x3F);
        }
        if ((extra == 1 && cp < 0x80) || (extra == 2 && cp < 0x800) || (extra == 3 && cp < 0x10000) || cp > 0x10FFFF ||
            (cp >= 0xD800 && cp <= 0xDFFF))
            return false;
        if (cp >= 0x10000) {
            cp -= 0x10000;
            esc(0xD800 + (cp >> 10));
            esc(0xDC00 + (cp & 0x3FF));
        } else esc(cp);
        p += extra + 1;
    }
    out += '"';
    return true;
}

bool put_float(std::string& out, double x) {              // repr(float); false: not finite
    char buf[40];
    const int n = tredbam_float_repr(x, buf);
    if (n < 0) return false;
    out.append(buf, (size_t)n);
    return true;
}

void put_int(std::string& out, long long v) {              // (by hand: a sample's text holds ~6 000 integers, snprintf was a fifth of the writer)
    char buf[24];
    char* p = buf + sizeof buf;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    out.append(p, (size_t)(buf + sizeof buf - p));
}

// "k|v;k|v" over a small histogram indexed by h (counter_s of tredparse/tred.py:149-150)
std::string counter_s(const std::vector<int32_t>& hist, int lo) {
    std::string s;
    for (size_t k = 0; k < hist.size(); ++k) {
        if (!hist[k]) continue;
        if (!s.empty()) s += ';';
        put_int(s, (long long)k + lo);
        s += '|';
        put_int(s, hist[k]);
    }
    return s;
}

struct PairStrings { std::string mean_std, hist; };
// mean_std ("346+/-78bp", population sd) and histogram ("0:0,25:3,...": 40 bins over [0, 1000]) of a pair-length list
// (models.py:87-98); "" for an empty list
PairStrings pair_strings(const int32_t* x, int32_t c) {
    PairStrings out;
    if (c <= 0) return out;
    constexpr int BINS = 40, SPAN = 1000, WIDTH = SPAN / BINS;
    int32_t h[BINS] = {};
    double sum = 0;
    for (int32_t i = 0; i < c; ++i) sum += (double)x[i];
    const double m = sum / (double)c;
    double q = 0;
    for (int32_t i = 0; i < c; ++i) {
        const double d = (double)x[i] - m;
        q += d * d;
        if (x[i] >= 0 && x[i] <= SPAN) ++h[std::min(x[i] / WIDTH, BINS - 1)];
    }
    char buf[96];
    out.mean_std.assign(buf, (size_t)snprintf(buf, sizeof buf, "%.0f+/-%.0fbp", m, std::sqrt(q / (double)c)));
    for (int j = 0; j < BINS; ++j) {
        if (j) out.hist += ',';
        put_int(out.hist, WIDTH * j);
        out.hist += ':';
        put_int(out.hist, h[j]);
    }
    return out;
}

// calc_label (models.py:370-392)
const char* label_of(const tredbam_emit_locus& t, int lo, int hi) {
    int decisive;
    bool at_risk;
    if (t.is_expansion) {
        decisive = t.is_recessive ? lo : hi;
        at_risk = decisive >= t.cutoff_risk;
    } else {
        decisive = t.is_recessive ? hi : lo;
        at_risk = 0 < decisive && decisive <= t.cutoff_risk;
    }
    if (t.cutoff_prerisk <= decisive && decisive < t.cutoff_risk) return "prerisk";
    if (at_risk) return "risk";
    return lo == -1 ? "missing" : "ok";
}

struct Entry { std::string key, text; };

// One deflate state per writer thread, reset between samples (setting one up allocates and clears 268 KB).  A sample's VCF text
// is 20 KB of which 18 are the records' REF / ALT alleles spelled out motif by motif: level 3 (no lazy matching) writes 4.6 KB
// where level 6 writes 4.1 KB, in half the time (0.18 against 0.35 ms per sample).
constexpr int GZIP_LEVEL = 3;

struct Gzipper {
    z_stream z;
    int level = 0;
    bool live = false;
    ~Gzipper() { if (live) deflateEnd(&z); }
    bool ready(int lv) {
        if (live && lv == level) return deflateReset(&z) == Z_OK;
        if (live) { deflateEnd(&z); live = false; }
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, lv, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        live = true;
        level = lv;
        return true;
    }
};

bool gzip_bytes(const std::string& text, int level, std::string& out) {
    static thread_local Gzipper g;
    if (!g.ready(level)) return false;
    z_stream& z = g.z;
    out.resize(deflateBound(&z, (uLong)text.size()) + 32);
    z.next_in = (Bytef*)text.data();
    z.avail_in = (uInt)text.size();
    z.next_out = (Bytef*)&out[0];
    z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    const size_t n = out.size() - z.avail_out;
    if (rc != Z_STREAM_END) { deflateEnd(&z); g.live = false; return false; }
    out.resize(n);
    return true;
}

bool write_file(const std::string& path, const char* data, size_t n) {
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) { g_emit_error = path + ": " + strerror(errno); return false; }
    while (n > 0) {
        const ssize_t w = write(fd, data, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            g_emit_error = path + ": " + strerror(errno);
            close(fd);
            return false;
        }
        data += w;
        n -= (size_t)w;
    }
    if (close(fd) != 0) { g_emit_error = path + ": " + strerror(errno); return false; }
    return true;
}

struct VcfLine { std::string chrom; int32_t pos; std::string line; };

}  // namespace

extern "C" {

const char* tredbam_emit_last_error(void) { return g_emit_error.c_str(); }

double tredbam_pairwise_sum(const double* a, int64_t n) { return (a && n > 0) ? 0.0 + pairwise(a, n) : 0.0; }

int tredbam_emit_sample_files(const tredbam_emit_locus* loci, int32_t n_loci, const tredbam_emit_batch* B,
                              const tredbam_emit_sample* S, const tredbam_emit_opts* O, int32_t* locus_status,
                              char* json_text, int64_t json_cap, int64_t* json_len) {
    if (!loci || n_loci < 0 || !B || !S || !O || !locus_status || !S->samplekey || !S->bam || !S->gender) return -2;
    if (json_len) *json_len = -1;
    for (int32_t k = 0; k < n_loci; ++k) locus_status[k] = 1;
    if (!S->opened) return 1;                       // (the Python path reports such a sample as the reference does)
    std::vector<Entry> entries;
    entries.reserve((size_t)n_loci * 22 + 3);
    auto add = [&](std::string key, std::string text) { entries.push_back(Entry{std::move(key), std::move(text)}); };
    auto quoted = [](const std::string& s) { return "\"" + s + "\""; };     // (strings built here: digits and ASCII punctuation)
    {
        std::string g;
        if (!json_string(g, S->gender)) return 1;
        add("inferredGender", g);
        std::string d;
        if (S->ydepth < 0) d = "-1";
        else if (!put_float(d, S->ydepth)) return 1;
        add("depthY", d);
        std::string r;
        put_int(r, S->readlen);
        add("readLen", r);
    }
    std::vector<VcfLine> vcf;
    std::vector<char> text_buf;
    std::vector<int64_t> det_reads;
    std::vector<uint8_t> det_tags;
    std::vector<int32_t> det_hs;
    for (int32_t k = 0; k < n_loci; ++k) {
        const int32_t u = S->unit_index ? S->unit_index[k] : -1;
        if (u < 0) continue;
        const tredbam_emit_locus& T = loci[k];
        const tredbam_emit_call& call = B->calls[u];
        if (call.status < 0) { locus_status[k] = call.status; continue; }
        const std::string n = T.name;
        // ---- format_call ----
        int a1 = -1, a2 = -1;
        std::string ci, pp_json = "-1", pp_vcf = "-1", p_h1 = "\"\"", p_h2 = "\"\"", p_joint = "\"\"";
        if (call.status != 1) {
            if (T.period <= 0 || call.h1 < 0 || call.h2 < 0) return 1;
            a1 = call.h1 / T.period;
            a2 = call.h2 / T.period;
            if (a1 > a2) std::swap(a1, a2);
            char buf[96];
            ci.assign(buf, (size_t)snprintf(buf, sizeof buf, "%d-%d|%d-%d", call.ci[0], call.ci[1], call.ci[2], call.ci[3]));
            pp_json.clear();
            if (!put_float(pp_json, call.pp)) return 1;
            pp_vcf.assign(buf, (size_t)snprintf(buf, sizeof buf, "%.4g", call.pp));
            // sparsify (models.py:304-317): entries >= e^-10, divided by the sum over the whole axis
            for (int side = 0; side < 2; ++side) {
                const double* P = B->marg + ((size_t)u * 2 + side) * (size_t)B->marg_len;
                const double total = 0.0 + pairwise(P, B->marg_len);
                std::vector<int32_t> keys;
                std::vector<double> vals;
                for (int64_t i = 0; i < B->marg_len; ++i)
                    if (P[i] >= SMALL_VALUE) { keys.push_back((int32_t)i); vals.push_back(P[i] / total); }
                text_buf.resize(64 + vals.size() * 96);
                const int64_t got = tredbam_sparse_json(keys.data(), nullptr, vals.data(), (int64_t)vals.size(), 2, text_buf.data(),
                                                        (int64_t)text_buf.size());
                if (got < 0) return got == -1 ? 1 : -2;
                (side ? p_h2 : p_h1).assign(text_buf.data(), (size_t)got);
            }
            {
                const int64_t lo = B->joint_lo[u], cnt = B->joint_n[u];
                std::vector<int32_t> ka((size_t)cnt), kb((size_t)cnt);
                for (int64_t i = 0; i < cnt; ++i) {
                    const int64_t x = B->joint_a[lo + i], y = B->joint_b[lo + i];
                    if (x < INT32_MIN || x > INT32_MAX || y < INT32_MIN || y > INT32_MAX) return 1;
                    ka[(size_t)i] = (int32_t)x;
                    kb[(size_t)i] = (int32_t)y;
                }
                text_buf.resize(64 + (size_t)cnt * 96);
                const int64_t got = tredbam_sparse_json(ka.data(), kb.data(), B->joint_v + lo, cnt, 2, text_buf.data(), (int64_t)text_buf.size());
                if (got < 0) return got == -1 ? 1 : -2;
                p_joint.assign(text_buf.data(), (size_t)got);
            }
        }
        const char* label = label_of(T, a1, a2);
        // ---- tally (bam_parser.py:174-182, 248-287) ----
        const tredbam_unit& U = S->unit[k];
        const int32_t r0 = B->unit_read_off[u], nr = B->unit_read_off[u + 1] - r0;
        if (nr != U.n_reads) return -2;
        const uint8_t* tags = B->tag + r0;
        const int16_t* hs = B->h + r0;
        det_reads.clear(); det_tags.clear(); det_hs.clear();
        for (int32_t i = 0; i < nr; ++i)
            if (tags[i] != TAG_NONE && tags[i] != TAG_HANG) {
                if (tags[i] > TAG_HANG) return 1;
                det_reads.push_back(i);
            }
        if (!B->repeatpairs && !det_reads.empty()) {
            // names tagged REPT twice or more: all their reads go (remove_pairs_of_rept)
            const int32_t* ids = S->name_id + U.read_first;
            int32_t top = 0;
            for (int64_t i : det_reads) top = std::max(top, ids[i]);
            std::vector<int32_t> rept((size_t)top + 1, 0);
            for (int64_t i : det_reads) if (tags[i] == TAG_REPT && ids[i] >= 0) ++rept[(size_t)ids[i]];
            size_t w = 0;
            for (int64_t i : det_reads) if (ids[i] < 0 || rept[(size_t)ids[i]] <= 1) det_reads[w++] = i;
            det_reads.resize(w);
        }
        int hmin = 0, hmax = 0;
        for (int64_t i : det_reads) { hmin = std::min(hmin, (int)hs[i]); hmax = std::max(hmax, (int)hs[i]); }
        std::vector<int32_t> full((size_t)(hmax - hmin + 1), 0), flank = full, rept = full;
        long long fdp = 0, pdp = 0, rdp = 0;
        for (int64_t& i : det_reads) {
            const int t = tags[i], h = hs[i] - hmin;
            if (t == TAG_FULL) { ++full[(size_t)h]; ++fdp; }
            else if (t == TAG_REPT) { ++rept[(size_t)h]; ++rdp; }
            else { ++flank[(size_t)h]; ++pdp; }
            det_tags.push_back((uint8_t)t);
            det_hs.push_back(hs[i]);
            i += U.read_first;                                     // (index into the scan's pools from here on)
        }
        std::string details;
        {
            int64_t name_bytes = 0, bases = 0;
            for (int64_t rd : det_reads) { name_bytes += S->name_off[rd + 1] - S->name_off[rd]; bases += S->read_len[rd]; }
            text_buf.resize((size_t)(64 + 200 * (int64_t)det_reads.size() + 2 * name_bytes + bases));
            const int64_t got = tredbam_details_json(S->seq4, S->seq4_off, S->read_len, S->names, S->name_off, det_reads.data(),
                                                     det_tags.data(), det_hs.data(), (int64_t)det_reads.size(), text_buf.data(),
                                                     (int64_t)text_buf.size());
            if (got < 0) return got == -1 ? 1 : -2;
            details.assign(text_buf.data(), (size_t)got);
        }
        const std::string fr = counter_s(full, hmin), pr = counter_s(flank, hmin), rr = counter_s(rept, hmin);
        std::string dp;
        if (!put_float(dp, S->depth[k])) return 1;
        const PairStrings g = pair_strings(S->global_lens ? S->global_lens + U.global_first : nullptr, U.n_global);
        const PairStrings t = pair_strings(S->target_lens ? S->target_lens + U.target_first : nullptr, U.n_target);
        auto num = [](long long v) { std::string s; put_int(s, v); return s; };
        add(n + ".1", num(a1));
        add(n + ".2", num(a2));
        add(n + ".FR", quoted(fr));
        add(n + ".PR", quoted(pr));
        add(n + ".RR", quoted(rr));
        add(n + ".DP", dp);
        add(n + ".FDP", num(fdp));
        add(n + ".PDP", num(pdp));
        add(n + ".RDP", num(rdp));
        add(n + ".PEDP", num(U.n_target));
        add(n + ".PEG", quoted(g.mean_std));
        add(n + ".PET", quoted(t.mean_std));
        add(n + ".P_PEG", quoted(g.hist));
        add(n + ".P_PET", quoted(t.hist));
        add(n + ".CI", quoted(ci));
        add(n + ".PP", pp_json);
        add(n + ".label", quoted(label));
        add(n + ".P_h1", std::move(p_h1));
        add(n + ".P_h2", std::move(p_h2));
        add(n + ".P_h1h2", std::move(p_joint));
        add(n + ".details", std::move(details));
        locus_status[k] = 0;
        // ---- the locus' VCF record (tred.py:316-374) ----
        if (O->write_vcf && T.in_vcf) {
            std::vector<int> novel;
            if (a1 != T.ref_copy) novel.push_back(a1);
            if (a2 != T.ref_copy && a2 != a1) novel.push_back(a2);       // (a1 <= a2: already sorted)
            std::string info = T.info, gt = "0/0", alt = ".";
            if (!novel.empty()) {
                info += ";RPA=";
                for (size_t i = 0; i < novel.size(); ++i) { if (i) info += ','; put_int(info, novel[i]); }
                gt = (a1 == T.ref_copy || a2 == T.ref_copy) ? "0/1" : (novel.size() == 1 ? "1/1" : "1/2");
                if (novel[0] != -1) {
                    alt.clear();
                    for (size_t i = 0; i < novel.size(); ++i) {
                        if (i) alt += ',';
                        for (int r = 0; r < novel[i]; ++r) alt += T.motif;
                    }
                }
            }
            std::string line = T.chrom;
            line += '\t'; put_int(line, T.pos);
            line += '\t'; line += T.name;
            line += '\t';
            for (int r = 0; r < T.ref_copy; ++r) line += T.motif;
            line += '\t'; line += alt;
            line += "\t.\t.\t"; line += info;
            line += "\tGT:GB:FR:PR:RR:DP:FDP:PDP:RDP:PEDP:CI:PP:LABEL\t";
            line += gt; line += ':'; put_int(line, a1); line += '/'; put_int(line, a2);
            line += ':'; line += fr; line += ':'; line += pr; line += ':'; line += rr; line += ':'; line += dp;
            line += ':'; put_int(line, fdp); line += ':'; put_int(line, pdp); line += ':'; put_int(line, rdp);
            line += ':'; put_int(line, U.n_target); line += ':'; line += ci; line += ':'; line += pp_vcf; line += ':'; line += label;
            vcf.push_back(VcfLine{T.chrom, T.pos, std::move(line)});
        }
    }
    // ---- the JSON text: sorted keys, 4-space indent ----
    std::sort(entries.begin(), entries.end(), [](const Entry& x, const Entry& y) { return x.key < y.key; });
    for (size_t i = 1; i < entries.size(); ++i)
        if (entries[i].key == entries[i - 1].key) return 1;          // (a locus listed twice: the dict path decides)
    size_t js_room = 256 + strlen(S->bam) * 2 + strlen(S->samplekey) * 2;
    for (const Entry& e : entries) js_room += e.key.size() + e.text.size() + 16;
    std::string js;
    js.reserve(js_room);                             // (540 KB per 30x sample: grown by doubling it was copied twice over)
    js = "{\n    \"bam\": ";
    if (!json_string(js, S->bam)) return 1;
    js += ",\n    \"samplekey\": ";
    if (!json_string(js, S->samplekey)) return 1;
    js += ",\n    \"tredCalls\": {\n";
    for (size_t i = 0; i < entries.size(); ++i) {
        js += "        ";
        if (!json_string(js, entries[i].key.c_str())) return 1;
        js += ": ";
        js += entries[i].text;
        js += i + 1 < entries.size() ? ",\n" : "\n";
    }
    js += "    }\n}";
    // ---- files: the VCF first, as write_vcf_json does ----
    const std::string key = S->samplekey;
    if (O->write_vcf) {
        std::string text = "##fileformat=VCFv4.1\n##fileDate=";
        text += O->filedate ? O->filedate : "";
        text += "\n##source="; text += O->source ? O->source : ""; text += ' '; text += S->bam;
        text += "\n##reference="; text += O->ref ? O->ref : "";
        text += "\n##inferredGender="; text += S->gender; text += " depthY=";
        if (S->ydepth < 0) text += "-1";
        else if (!put_float(text, S->ydepth)) return 1;
        text += "\n##readLen="; put_int(text, S->readlen); text += "bp\n";
        text += O->vcf_meta ? O->vcf_meta : "";
        text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t";
        text += key;
        text += '\n';
        std::sort(vcf.begin(), vcf.end(), [](const VcfLine& x, const VcfLine& y) {
            if (x.chrom != y.chrom) return x.chrom < y.chrom;
            if (x.pos != y.pos) return x.pos < y.pos;
            return x.line < y.line;
        });
        for (const VcfLine& l : vcf) { text += l.line; text += '\n'; }
        std::string gz;
        if (!gzip_bytes(text, O->gzip_level > 0 ? O->gzip_level : GZIP_LEVEL, gz)) { g_emit_error = "gzip failed"; return -5; }
        if (!write_file(key + ".tred.vcf.gz", gz.data(), gz.size())) return -5;
    }
    if (O->write_json) {
        js += '\n';
        const bool ok = write_file(key + ".json", js.data(), js.size());
        js.pop_back();
        if (!ok) return -5;
    }
    if (json_text && json_len) {
        if ((int64_t)js.size() <= json_cap) { memcpy(json_text, js.data(), js.size()); *json_len = (int64_t)js.size(); }
        else *json_len = -(int64_t)js.size() - 1;
    }
    return 0;
}

}  // extern "C"
