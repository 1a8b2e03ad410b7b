// inflate_block.h -- DEFLATE (RFC 1951) decoder for whole BGZF blocks, host only.
//
// A BGZF block is a complete raw-deflate stream of at most 64 KiB of output whose compressed bytes and output size
// are both known before decoding starts.  That allows a decoder without any streaming state: a 64-bit bit buffer
// refilled eight bytes at a time, one table lookup per symbol (11-bit primary table for literals/lengths, 8-bit for
// distances, second-level tables for the rare longer codes), the extra bits of a length or distance taken from the
// same refill as its code, and matches copied eight bytes at a time into an output buffer with slack at its end.
// On the synthetic 30x BAMs inflate was ~70 % of tredbam_scan with zlib 1.2.11; this decoder is ~2x zlib there.
// Written from RFC 1951; no third-party code.  Returns false on anything it does not like (the caller then falls
// back to zlib, which also produces the error message for truly corrupt input).
#ifndef TREDBAM_INFLATE_BLOCK_H
#define TREDBAM_INFLATE_BLOCK_H

#include <cstdint>
#include <cstring>

namespace tredbam_inflate {

constexpr int LL_BITS = 11;       // primary table bits, literal/length alphabet
constexpr int D_BITS = 8;         // primary table bits, distance alphabet
constexpr int LL_ENTRIES = (1 << LL_BITS) + 2048;   // + second-level tables (15 - 11 bits deep, <= 288 symbols)
constexpr int D_ENTRIES = (1 << D_BITS) + 512;
constexpr int SLACK = 32;         // bytes the output buffer must have beyond the expected size

// table entry: bits 0-4 code length to consume (second-level entries: the bits left after the primary ones);
// bits 5-8 number of extra bits (pointer entries: bits of the second-level index); bits 9-12 flags;
// bits 16-31 value (literal, base length, base distance, start of the second-level table)
constexpr uint32_t F_LITERAL = 1u << 9, F_SUB = 1u << 10, F_EOB = 1u << 11, F_BAD = 1u << 12;
constexpr uint32_t F_LIT2 = 1u << 13;   // literal entry that carries a second literal in bits 24-31 (length field: both codes)
inline int e_len(uint32_t e) { return (int)(e & 31); }
inline int e_extra(uint32_t e) { return (int)((e >> 5) & 15); }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

struct Tables {
    uint32_t ll[LL_ENTRIES];
    uint32_t d[D_ENTRIES];
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                      131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                       2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bitrev(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// value / flags of symbol `sym` of the literal-length (ll = true) or distance alphabet, without the length field
inline uint32_t symbol_entry(bool ll, int sym) {
    if (ll) {
        if (sym < 256) return F_LITERAL | ((uint32_t)sym << 16);
        if (sym == 256) return F_EOB;
        if (sym > 285) return F_BAD;
        return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 5);
    }
    if (sym > 29) return F_BAD;
    return ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 5);
}

// Canonical Huffman decoding table from code lengths (RFC 1951 3.2.2).  false: over-subscribed code or table overflow.
// An incomplete code is accepted (its unused slots decode as F_BAD), as zlib accepts a single-code distance tree.
inline bool build(const uint8_t* lens, int n, bool ll, uint32_t* table, int table_cap) {
    const int tbits = ll ? LL_BITS : D_BITS;
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    for (int i = 0; i < (1 << tbits); ++i) table[i] = F_BAD | 1u;      // consumes a bit, flagged: invalid code
    // longest code behind every primary prefix that needs a second level
    uint8_t deepest[1 << LL_BITS];
    memset(deepest, 0, (size_t)1 << tbits);
    uint32_t codes[288];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = bitrev(next_code[l]++, l);
        codes[s] = r;
        if (l > tbits) {
            const uint32_t p = r & ((1u << tbits) - 1);
            if (l > deepest[p]) deepest[p] = (uint8_t)l;
        }
    }
    int used = 1 << tbits;
    for (int p = 0; p < (1 << tbits); ++p) {
        if (!deepest[p]) continue;
        const int sb = deepest[p] - tbits;
        if (used + (1 << sb) > table_cap) return false;
        table[p] = F_SUB | ((uint32_t)used << 16) | ((uint32_t)sb << 5) | (uint32_t)tbits;
        for (int i = 0; i < (1 << sb); ++i) table[used + i] = F_BAD | 1u;
        used += 1 << sb;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = codes[s];
        const uint32_t e = symbol_entry(ll, s);
        if (l <= tbits) {
            for (uint32_t i = r; i < (1u << tbits); i += 1u << l) table[i] = e | (uint32_t)l;
        } else {
            const uint32_t pe = table[r & ((1u << tbits) - 1)];
            const int sb = e_extra(pe);
            const uint32_t start = e_value(pe);
            const int rest = l - tbits;
            for (uint32_t i = r >> tbits; i < (1u << sb); i += 1u << rest) table[start + i] = e | (uint32_t)rest;
        }
    }
    if (ll) {
        // Two literals per look-up where both codes fit the primary index (4-bit sequence bytes and binned qualities
        // get 4-6 bit codes): the entry keeps the first literal in bits 16-23, takes the second into bits 24-31 and
        // counts both codes in its length.  The second code is read off the single-literal table: index = the bits left
        // after the first code, upper bits zero -- valid iff that entry is a literal no longer than those bits.
        // (Descending order: the entries looked up, at indices i >> l1 < i, are still single.)
        for (int i = (1 << tbits) - 1; i >= 0; --i) {
            const uint32_t e1 = table[i];
            if (!(e1 & F_LITERAL)) continue;
            const int l1 = e_len(e1);
            const uint32_t e2 = table[(uint32_t)i >> l1];
            if ((e2 & (F_LITERAL | F_LIT2)) != F_LITERAL || e_len(e2) > tbits - l1) continue;
            table[i] = F_LITERAL | F_LIT2 | (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8) | (uint32_t)(l1 + e_len(e2));
        }
    }
    return true;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int n = 0;        // valid bits in buf
    // after refill at least 56 bits are valid while input remains; past the end zeros are shifted in and `over`
    // counts them (a well-formed stream never consumes them)
    int over = 0;
    inline void refill() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << n;
            const int take = (63 - n) >> 3;
            p += take;
            n += take << 3;
        } else {
            while (n <= 56) {
                if (p < end) buf |= (uint64_t)*p++ << n;
                else ++over;
                n += 8;
            }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; n -= k; }
    inline bool overrun() const { return n < 8 * over; }       // bits beyond the end of the input were consumed
};

// Inflate one raw-deflate stream of exactly out_len bytes into out (which has SLACK writable bytes beyond out_len).
// (always inlined: the caller's target options -- bamread.cpp compiles it once per CPU clone -- apply to the loop)
__attribute__((always_inline)) inline bool inflate_block(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, Tables& T) {
    Bits b;
    b.p = in;
    b.end = in + in_len;
    uint8_t* o = out;
    uint8_t* const oend = out + out_len;
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        b.refill();
        const int final_block = (int)b.peek(1);
        const int type = (int)((b.buf >> 1) & 3);
        b.drop(3);
        if (type == 0) {
            b.drop(b.n & 7);                      // to the byte boundary
            if (b.overrun()) return false;
            // unread whole bytes still in the bit buffer belong to the stream (padding past its end does not)
            const uint8_t* q = b.p - ((b.n - 8 * b.over) >> 3);
            if (b.end - q < 4) return false;
            const uint32_t len = q[0] | ((uint32_t)q[1] << 8), nlen = q[2] | ((uint32_t)q[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            q += 4;
            if ((size_t)(b.end - q) < len || (size_t)(oend - o) < len) return false;
            memcpy(o, q, len);
            o += len;
            b.p = q + len;
            b.buf = 0;
            b.n = 0;
            b.over = 0;
        } else if (type == 1 || type == 2) {
            uint8_t lens[320];
            int hlit, hdist;
            if (type == 1) {
                hlit = 288; hdist = 30;
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                for (int i = 0; i < 30; ++i) lens[288 + i] = 5;
            } else {
                hlit = (int)b.peek(5) + 257; b.drop(5);
                hdist = (int)b.peek(5) + 1; b.drop(5);
                const int hclen = (int)b.peek(4) + 4; b.drop(4);
                if (hlit > 286 || hdist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; ++i) {
                    if (b.n < 3) b.refill();
                    cl[ORDER[i]] = (uint8_t)b.peek(3);
                    b.drop(3);
                }
                uint32_t ct[1 << 7];
                {   // code-length code: at most 7 bits, a flat 128-entry table
                    int count[8] = {0};
                    for (int i = 0; i < 19; ++i) ++count[cl[i]];
                    count[0] = 0;
                    int left = 1;
                    for (int l = 1; l <= 7; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
                    uint32_t nc[8], code = 0;
                    for (int l = 1; l <= 7; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; nc[l] = code; }
                    for (int i = 0; i < 128; ++i) ct[i] = 0xFFFFFFFFu;
                    for (int s = 0; s < 19; ++s) {
                        const int l = cl[s];
                        if (!l) continue;
                        const uint32_t r = bitrev(nc[l]++, l);
                        for (uint32_t i = r; i < 128; i += 1u << l) ct[i] = ((uint32_t)s << 8) | (uint32_t)l;
                    }
                }
                int i = 0;
                while (i < hlit + hdist) {
                    if (b.n < 14) b.refill();
                    const uint32_t e = ct[b.peek(7)];
                    if (e == 0xFFFFFFFFu) return false;
                    b.drop((int)(e & 255));
                    const int sym = (int)(e >> 8);   // (the flat code-length table keeps its own simple layout)
                    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) {
                        if (i == 0) return false;
                        v = lens[i - 1];
                        rep = 3 + (int)b.peek(2); b.drop(2);
                    } else if (sym == 17) { rep = 3 + (int)b.peek(3); b.drop(3); }
                    else { rep = 11 + (int)b.peek(7); b.drop(7); }
                    if (i + rep > hlit + hdist) return false;
                    while (rep--) lens[i++] = v;
                }
                if (lens[256] == 0) return false;      // no end-of-block code
                // distance lengths follow the literal/length ones directly: move them to a fixed place
                memmove(lens + 288, lens + hlit, (size_t)hdist);
            }
            if (!build(lens, hlit, true, T.ll, LL_ENTRIES)) return false;
            if (!build(lens + 288, hdist, false, T.d, D_ENTRIES)) return false;
            // ---- symbols, fast loop: while 16 input bytes and the longest match plus the copy slack are in reach no
            //      bound needs testing except a match's distance; the careful loop below finishes the block.
            //      Every path ends with "refill, look the next symbol up": the table load of the next symbol is in
            //      flight while a match is copied (BAM blocks are match-dominated: ~12 bytes per match), and the bit
            //      buffer holds >= 56 bits whenever a symbol is taken apart (length 15 + 5, distance 15 + 13). ----
            bool eob = false;
#define TREDBAM_REFILL { uint64_t w; memcpy(&w, b.p, 8); b.buf |= w << b.n; const int take = (63 - b.n) >> 3; b.p += take; b.n += take << 3; }
            if (b.end - b.p >= 16) {
                TREDBAM_REFILL
                uint32_t e = T.ll[b.peek(LL_BITS)];
                while (b.end - b.p >= 16 && oend - o >= 258 + 32) {
                    if (e & F_LITERAL) {                      // up to three look-ups per refill (3 x 15 <= 56 bits),
                                                              // each one or two literals (the second byte written for a
                                                              // single literal is scratch: the output has room here)
#define TREDBAM_PUT_LITERALS(E) { b.drop(e_len(E)); o[0] = (uint8_t)((E) >> 16); o[1] = (uint8_t)((E) >> 24); o += 1 + (((E) >> 13) & 1u); }
                        TREDBAM_PUT_LITERALS(e)
                        e = T.ll[b.peek(LL_BITS)];
                        if (e & F_LITERAL) {
                            TREDBAM_PUT_LITERALS(e)
                            e = T.ll[b.peek(LL_BITS)];
                            if (e & F_LITERAL) {
                                TREDBAM_PUT_LITERALS(e)
                                e = T.ll[b.peek(LL_BITS)];        // (>= 56 - 45 = 11 bits are left: enough for the index)
                                TREDBAM_REFILL
                                continue;
                            }
                        }
#undef TREDBAM_PUT_LITERALS
                        TREDBAM_REFILL                        // e stays valid: a refill only adds bits above the ones seen
                        continue;
                    }
                    if (e & F_SUB) {
                        b.drop(LL_BITS);
                        e = T.ll[e_value(e) + b.peek(e_extra(e))];
                        if (e & F_LITERAL) {
                            b.drop(e_len(e));
                            *o++ = (uint8_t)e_value(e);
                            TREDBAM_REFILL
                            e = T.ll[b.peek(LL_BITS)];
                            continue;
                        }
                    }
                    if (e & (F_EOB | F_BAD)) {
                        if (e & F_BAD) return false;
                        b.drop(e_len(e));
                        eob = true;
                        break;
                    }
                    const int cl = e_len(e), xb = e_extra(e);
                    const uint32_t len = e_value(e) + ((uint32_t)(b.buf >> cl) & ((1u << xb) - 1));
                    b.drop(cl + xb);
                    uint32_t de = T.d[b.peek(D_BITS)];
                    if (de & F_SUB) {
                        b.drop(D_BITS);
                        de = T.d[e_value(de) + b.peek(e_extra(de))];
                    }
                    if (de & F_BAD) return false;
                    const int dcl = e_len(de), dxb = e_extra(de);
                    const uint32_t dist = e_value(de) + ((uint32_t)(b.buf >> dcl) & ((1u << dxb) - 1));
                    b.drop(dcl + dxb);
                    // the next symbol's entry, loaded while the match is copied -- from the bits at hand when they cover
                    // the index (nearly always), so that the refill is off the chain load -> shift -> load -> shift
                    if (b.n >= LL_BITS) {
                        e = T.ll[b.peek(LL_BITS)];
                        TREDBAM_REFILL
                    } else {
                        TREDBAM_REFILL
                        e = T.ll[b.peek(LL_BITS)];
                    }
                    if (dist > (size_t)(o - out)) return false;
                    const uint8_t* s = o - dist;
                    uint8_t* const stop = o + len;
                    if (dist >= 16) {
                        memcpy(o, s, 16);                                              // most matches are short
                        if (len > 16) { memcpy(o + 16, s + 16, 16); if (len > 32) { o += 32; s += 32; do { memcpy(o, s, 16); o += 16; s += 16; } while (o < stop); } }
                    } else if (dist >= 8) {
                        memcpy(o, s, 8); memcpy(o + 8, s + 8, 8);
                        if (len > 16) { o += 16; s += 16; do { memcpy(o, s, 8); o += 8; s += 8; } while (o < stop); }
                    } else if (dist == 1) {
                        memset(o, *s, len);
                    } else {
                        do { *o++ = *s++; } while (o < stop);
                    }
                    o = stop;
                }
            }
#undef TREDBAM_REFILL
            // ---- symbols, careful loop ----
            while (!eob) {
                b.refill();                                   // >= 56 bits: a length (15 + 5) and a distance (15 + 13) fit
                uint32_t e = T.ll[b.peek(LL_BITS)];
                if (e & F_SUB) {
                    b.drop(LL_BITS);
                    e = T.ll[e_value(e) + b.peek(e_extra(e))];
                }
                if (e & F_LITERAL) {
                    b.drop(e_len(e));
                    const int cnt = 1 + (int)((e >> 13) & 1u);        // (second-level entries are always single)
                    if (oend - o < cnt) return false;
                    o[0] = (uint8_t)(e >> 16);
                    if (cnt == 2) o[1] = (uint8_t)(e >> 24);
                    o += cnt;
                    continue;
                }
                if (e & (F_EOB | F_BAD)) {
                    if (e & F_BAD) return false;
                    b.drop(e_len(e));
                    break;
                }
                const int cl = e_len(e), xb = e_extra(e);
                const uint32_t len = e_value(e) + ((uint32_t)(b.buf >> cl) & ((1u << xb) - 1));
                b.drop(cl + xb);
                uint32_t de = T.d[b.peek(D_BITS)];
                if (de & F_SUB) {
                    b.drop(D_BITS);
                    de = T.d[e_value(de) + b.peek(e_extra(de))];
                }
                if (de & F_BAD) return false;
                const int dcl = e_len(de), dxb = e_extra(de);
                const uint32_t dist = e_value(de) + ((uint32_t)(b.buf >> dcl) & ((1u << dxb) - 1));
                b.drop(dcl + dxb);
                if (dist > (size_t)(o - out) || len > (size_t)(oend - o)) return false;
                const uint8_t* s = o - dist;
                uint8_t* const stop = o + len;
                if (dist >= 8) {
                    do { memcpy(o, s, 8); o += 8; s += 8; } while (o < stop);     // may run up to 7 bytes past: SLACK
                } else if (dist == 1) {
                    memset(o, *s, len);
                } else {
                    do { *o++ = *s++; } while (o < stop);
                }
                o = stop;
            }
        } else {
            return false;
        }
        if (b.overrun()) return false;
        if (final_block) break;
    }
    return o == oend;
}

}  // namespace tredbam_inflate
#endif
