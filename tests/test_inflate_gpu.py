"""GPU parity: the batch DEFLATE decoder (tredgpu_inflate_blocks) against zlib, byte for byte -- stored, fixed and
dynamic blocks, every compression level, several deflate blocks per stream, the repository's BAM fixtures block by
block, empty streams -- and its status codes on damaged streams."""
import os
import struct
import zlib

import numpy as np
import pytest

from tredparse_amd import _lib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), flush_every):
        out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)      # several deflate blocks
    return out + c.flush()


def _inflate(inf, payloads, sizes, crc=False):
    n = len(payloads)
    offs = np.zeros(n + 1, np.int64)
    for k, p in enumerate(payloads):
        offs[k + 1] = (offs[k] + len(p) + 3) & ~3
    comp, out, coff, ooff = inf.reserve(int(offs[-1]), int(sum(sizes)), n)
    comp[:] = 0
    for k, p in enumerate(payloads):
        comp[offs[k]:offs[k] + len(p)] = np.frombuffer(p, np.uint8)
    # (the decoder takes a payload's end from the next payload's start: ends are rounded up to the 4-byte grid, and a
    #  stream that needs bytes beyond its own last one is caught by zlib-style checks on the host side, CRC included)
    coff[:] = offs
    ooff[0] = 0
    ooff[1:] = np.cumsum(sizes)
    if crc:
        status, sums = inf.run(n, crc=True)
        return status, [bytes(out[ooff[k]:ooff[k + 1]]) for k in range(n)], sums
    status = inf.run(n)
    return status, [bytes(out[ooff[k]:ooff[k + 1]]) for k in range(n)]


@pytest.fixture(scope="module")
def inf():
    f = _lib.Inflater(0)
    yield f
    f.close()


def _bam_like(rng, n):
    rec = bytearray()
    while len(rec) < n:
        name = b"read%07d" % int(rng.integers(10 ** 7))
        seq = bytes(rng.integers(0, 256, 75, dtype=np.uint8))
        qual = bytes(rng.choice(np.array([2, 11, 25, 37], np.uint8), 150))
        rec += struct.pack("<iiiIIiii", 300, 3, int(rng.integers(1 << 27)), 0x12345678, 0x0990000, 150, 3, 0) + name + b"\0" + seq + qual
    return bytes(rec[:n])


def test_random_streams_match_zlib(inf):
    rng = np.random.default_rng(11)
    datas, payloads = [], []
    for k in range(300):
        n = int(rng.choice([0, 1, 2, 100, 4000, 30000, 65280, 65536]))
        kind = k % 5
        if kind == 0:
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))                      # incompressible
        elif kind == 1:
            d = bytes(rng.integers(0, 4, n, dtype=np.uint8))                        # few symbols, long codes elsewhere
        elif kind == 2:
            d = (b"CAG" * (n // 3 + 1))[:n]                                         # overlapping matches, distance 3
        elif kind == 3:
            d = _bam_like(rng, n)
        else:
            d = bytes(np.repeat(rng.integers(0, 256, max(n // 300, 1), dtype=np.uint8), 300)[:n])   # runs: distance 1
        level = int(rng.choice([0, 1, 4, 6, 9]))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE][int(rng.integers(4))]
        flush = int(rng.choice([0, 0, 5000]))
        datas.append(d)
        payloads.append(_raw(d, level, strategy, flush))
    status, got = _inflate(inf, payloads, [len(d) for d in datas])
    assert (status == 0).all(), np.nonzero(status)[0][:10]
    for k, d in enumerate(datas):
        assert got[k] == d, k


def _bgzf_blocks(path):
    raw = open(path, "rb").read()
    pos, blocks = 0, []
    while pos < len(raw):
        xlen = struct.unpack_from("<H", raw, pos + 10)[0]
        bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1          # (the fixtures carry BC as their first extra field)
        payload = raw[pos + 12 + xlen:pos + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, pos + bsize - 8)
        blocks.append((payload, crc, isize))
        pos += bsize
    return blocks


@pytest.mark.parametrize("name", ["t001.bam", "t002.bam", "synf.bam"])
def test_every_block_of_the_bam_fixtures(inf, name):
    blocks = _bgzf_blocks(os.path.join(GOLD, "bam", name))
    status, got = _inflate(inf, [b[0] for b in blocks], [b[2] for b in blocks])
    assert (status == 0).all()
    for (payload, crc, isize), data in zip(blocks, got):
        assert len(data) == isize and zlib.crc32(data) == crc
        assert data == zlib.decompress(payload, -15)


def test_damaged_streams_are_reported_per_block(inf):
    rng = np.random.default_rng(3)
    good = _bam_like(rng, 20000)
    p = _raw(good)
    cases = [p, p[:len(p) // 2], b"\x07" + p[1:], p, bytes(8)]     # ok, truncated, reserved block type, wrong size, stored with bad LEN
    sizes = [len(good), len(good), len(good), len(good) + 5, 10]
    status, got = _inflate(inf, cases, sizes)
    assert status[0] == 0 and got[0] == good
    assert status[1] != 0 and status[2] == -1 and status[3] == -2 and status[4] == -1
    # a second call on the same inflater is clean again
    status, got = _inflate(inf, [p], [len(good)])
    assert status[0] == 0 and got[0] == good


def test_many_blocks_in_one_call(inf):
    rng = np.random.default_rng(5)
    datas = [_bam_like(rng, 65280) for _ in range(40)]
    payloads = [_raw(d) for d in datas] * 25                      # 1 000 blocks, 65 MB of output
    status, got = _inflate(inf, payloads, [65280] * len(payloads))
    assert (status == 0).all()
    assert all(got[k] == datas[k % 40] for k in range(0, len(payloads), 37))
    # 9 000 blocks: the call is cut into slices that alternate between two streams (copy-out of one slice beside the
    # decoding of the next); the device CRCs say every block arrived as decoded
    payloads = payloads * 9
    want = [zlib.crc32(d) for d in datas]
    status, got, sums = _inflate(inf, payloads, [65280] * len(payloads), crc=True)
    assert (status == 0).all() and [int(c) for c in sums] == [want[k % 40] for k in range(len(payloads))]
    assert all(got[k] == datas[k % 40] for k in range(0, len(payloads), 211))
    total, kernel = inf.timing()
    assert 0 < kernel <= total * 2 + 1                            # (two streams: the decode launches overlap each other)


def test_corrupted_streams_never_write_or_read_out_of_bounds(inf):
    """Bit flips, truncations and garbage -- also in the last payload of the staging buffer, where a stream that keeps
    asking for bits would run off the end: every block comes back with a status, blocks that zlib still inflates to the
    expected size are byte-identical, the neighbours of a damaged block are untouched."""
    rng = np.random.default_rng(17)
    good = [_bam_like(rng, int(rng.choice([3000, 20000, 65280]))) for _ in range(24)]
    payloads, sizes, want = [], [], []
    for k in range(400):
        d = good[k % len(good)]
        p = bytearray(_raw(d, int(rng.choice([1, 6, 9]))))
        kind = k % 5
        if kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                p[int(rng.integers(len(p)))] ^= 1 << int(rng.integers(8))
        elif kind == 2:
            p = p[:int(rng.integers(1, len(p)))]
        elif kind == 3:
            p = bytearray(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes())
        payloads.append(bytes(p))
        sizes.append(len(d))
        try:
            got = zlib.decompressobj(-15).decompress(bytes(p))
            want.append(got if len(got) == len(d) else None)
        except zlib.error:
            want.append(None)
    for last in (2, 3, 1):                                    # a truncated / garbage / bit-flipped stream last in the buffer
        order = [k for k in range(400) if k % 5 != last] + [k for k in range(400) if k % 5 == last]
        status, got = _inflate(inf, [payloads[k] for k in order], [sizes[k] for k in order])
        for j, k in enumerate(order):
            if k % 5 in (0, 4):
                assert status[j] == 0 and got[j] == good[k % len(good)], k
            elif status[j] == 0 and want[k] is not None:
                assert got[j] == want[k], k                   # (a flip the stream survives: same bytes as zlib's)
            elif want[k] is None and status[j] == 0:
                # zlib refuses what the decoder took: only possible where zlib asks for more than a valid prefix
                # (trailing garbage / a missing end); the host's CRC-32 check is what catches these
                assert len(got[j]) == sizes[k]


def test_device_crc_matches_zlib(inf):
    """The CRC-32 the decoding wavefront computes over its own output (tredgpu_inflate_blocks_crc) is zlib's, for every
    size class of the chunking (64 equal power-of-two chunks padded in front), and 0 for blocks that failed."""
    rng = np.random.default_rng(23)
    sizes = [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 16384, 30000, 61440, 61441, 65279, 65280, 65535, 65536]
    datas = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) if k % 2 else _bam_like(rng, n) for k, n in enumerate(sizes * 3)]
    payloads = [_raw(d, int(rng.choice([1, 6]))) for d in datas]
    payloads.append(payloads[-1][:len(payloads[-1]) // 2])          # a damaged one: no checksum
    datas.append(datas[-1])
    status, got, sums = _inflate(inf, payloads, [len(d) for d in datas], crc=True)
    assert (status[:-1] == 0).all() and status[-1] != 0 and sums[-1] == 0
    for k, d in enumerate(datas[:-1]):
        assert got[k] == d and int(sums[k]) == zlib.crc32(d), (k, len(d))
    blocks = _bgzf_blocks(os.path.join(GOLD, "bam", "t001.bam"))
    status, got, sums = _inflate(inf, [b[0] for b in blocks], [b[2] for b in blocks], crc=True)
    assert (status == 0).all() and [int(c) for c in sums] == [b[1] for b in blocks]
    total, kernel = inf.timing()
    assert 0 < kernel <= total


# ---- hand-made dynamic blocks: code length sets that need the deepest second-level tables ------------------------------------
class _Bits(object):
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, nbits):                 # value's low bit first (header fields, extra bits)
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):                 # a Huffman code: its first (most significant) bit first
        for i in range(nbits - 1, -1, -1):
            self.put((code >> i) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def _canonical(lens):
    count = [0] * 17
    for l in lens:
        count[l] += 1
    count[0] = 0
    nxt, code = [0] * 17, 0
    for l in range(1, 16):
        code = (code + count[l - 1]) << 1
        nxt[l] = code
    codes = []
    for l in lens:
        codes.append(nxt[l] if l else 0)
        nxt[l] += 1 if l else 0
    return codes


LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0] + [k // 2 for k in range(2, 28)]


def _dynamic_block(l_lens, d_lens, symbols):
    """One final dynamic block with the given code lengths (286 + 30, each sent as a plain 4-bit code-length code) and
    symbols: an int = a literal, (length code 0..28, its extra bits, distance code 0..29, its extra bits) = a match."""
    b = _Bits()
    b.put(1, 1); b.put(2, 2)
    b.put(len(l_lens) - 257, 5); b.put(len(d_lens) - 1, 5); b.put(19 - 4, 4)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    for s in order:
        b.put(4 if s < 16 else 0, 3)             # code lengths 0..15 as themselves: sixteen 4-bit codes, complete
    for l in list(l_lens) + list(d_lens):
        b.code(l, 4)
    lc, dc = _canonical(l_lens), _canonical(d_lens)
    for s in symbols:
        if isinstance(s, int):
            b.code(lc[s], l_lens[s])
        else:
            c, cx, d, dx = s
            b.code(lc[257 + c], l_lens[257 + c]); b.put(cx, LEN_EXTRA[c])
            b.code(dc[d], d_lens[d]); b.put(dx, DIST_EXTRA[d])
    b.code(lc[256], l_lens[256])
    return b.done()


# the code length sets that need the most second-level table entries behind a 9-bit / 7-bit root (a hill climb over all
# complete codes of 286 / 30 symbols up to 15 bits: 338 and 272 entries of the decoder's 640)
WORST_L = [1, 2, 3, 4, 7] + [10] * 11 + [11] * 41 + [12] * 49 + [13] * 49 + [14] * 65 + [15] * 66
WORST_D = [1, 2, 3, 4, 5, 8] + [9] * 9 + [10] * 9 + [11, 12, 13, 14, 15, 15]


def _complete(lens):
    return sum(2 ** (15 - l) for l in lens) == 2 ** 15


@pytest.mark.parametrize("shuffle", [0, 1, 2])
def test_codes_that_need_the_deepest_second_level_tables(inf, shuffle):
    """Every code longer than the root tables' 9 / 7 bits is found through a link to a second-level table; here the
    literal / length and the distance code are the complete codes that need the most of them, every symbol of both is
    used, and zlib has to read the same bytes."""
    rng = np.random.default_rng(40 + shuffle)
    l_lens, d_lens = list(WORST_L), list(WORST_D)
    assert len(l_lens) == 286 and len(d_lens) == 30 and _complete(l_lens) and _complete(d_lens)
    if shuffle:
        rng.shuffle(l_lens); rng.shuffle(d_lens)
    symbols = [int(x) for x in rng.permutation(256)] * 3                    # 768 bytes first: room for every short distance
    produced = len(symbols)
    for c in rng.permutation(29):
        for d in rng.permutation(30):
            dx = int(rng.integers(1 << DIST_EXTRA[d])) if DIST_EXTRA[d] else 0
            if DIST_BASE[d] + dx > produced:
                continue
            cx = int(rng.integers(1 << LEN_EXTRA[c])) if LEN_EXTRA[c] else 0
            symbols.append((int(c), cx, int(d), dx))
            produced += LEN_BASE[c] + cx
            symbols.append(int(rng.integers(256)))
            produced += 1
            if produced > 60000:
                break
        if produced > 60000:
            break
    stream = _dynamic_block(l_lens, d_lens, symbols)
    want = zlib.decompressobj(-15).decompress(stream)
    assert len(want) == produced
    status, got = _inflate(inf, [stream, _raw(want)], [len(want), len(want)])
    assert list(status) == [0, 0] and got[0] == want and got[1] == want
