"""CPU: the BGZF block decoder of libtredbam.so (csrc/inflate_block.h, written from RFC 1951) against zlib -- every
block type (stored, fixed, dynamic), several deflate blocks per stream, inputs around the fast loop's margins, every
block of the test BAMs; corrupt, truncated or wrongly sized streams are declined (the reader then asks zlib), never
decoded wrongly and never crash."""
import ctypes as C
import os
import random
import struct
import zlib

import pytest

from tredparse_amd import bamio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lib():
    lib = bamio._native()
    assert lib is not None, "libtredbam.so is not built"
    lib.tredbam_inflate_raw.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
    lib.tredbam_inflate_raw.restype = C.c_int
    return lib


def _bam_like(rng, n_bytes):
    recs = []
    for i in range(n_bytes // 290 + 1):
        recs.append(struct.pack("<iiBBHHHiiii", 3, 1000 + i * 7, 20, 60, 4681, 1, 99, 150, 3, 1200 + i * 7, 350)
                    + ("r.%07d" % i).encode() + b"\0" + struct.pack("<I", 150 << 4) + os.urandom(75) + b"\xff" * 150)
    return b"".join(recs)[:n_bytes]


def _payload(rng, kind, n):
    if kind == 0: return os.urandom(n)
    if kind == 1: return bytes(rng.choice(b"ACGT") for _ in range(n))
    if kind == 2: return (b"CAG" * (n // 3 + 1))[:n]
    if kind == 3: return bytes(n)
    if kind == 4:
        base = os.urandom(rng.randrange(1, 300))
        return (base * (n // len(base) + 1))[:n]
    if kind == 5: return _bam_like(rng, n)
    return bytes(min(255, int(rng.expovariate(0.05))) for _ in range(n))


def test_streams_of_every_kind_decode_like_zlib(lib):
    rng = random.Random(2)
    for trial in range(1500):
        n = rng.choice([0, 1, 2, 3, 10, 100, 273, 274, 275, 300, 1000, 5000, 65280, 65536]) if rng.random() < 0.3 \
            else rng.randrange(0, 65537)
        data = _payload(rng, rng.randrange(7), n)
        level = rng.choice([0, 1, 1, 6, 9])
        strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 3 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 8, 9]), strategy)
        parts, prev = [], 0
        if rng.random() < 0.3:          # several deflate blocks in one stream (a flush also emits an empty stored block)
            for c in sorted(rng.sample(range(len(data) + 1), min(3, len(data) + 1))):
                parts += [co.compress(data[prev:c]), co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))]
                prev = c
        parts += [co.compress(data[prev:]), co.flush()]
        comp = b"".join(parts)
        out = C.create_string_buffer(max(len(data), 1))
        assert lib.tredbam_inflate_raw(comp, len(comp), out, len(data)) == 1, (trial, n, level, strategy)
        assert out.raw[:len(data)] == data, (trial, n, level, strategy)


def test_every_block_of_the_test_bams(lib, tmp_path):
    from tredparse_amd import synth, synth_bam
    recs, _ = synth_bam.simulate_sample(4, [l for l in synth.load_loci() if l["name"] in ("HD", "DM1")])
    syn = str(tmp_path / "s.bam")
    synth_bam.write_bam(syn, recs, level=6)
    n_blocks = 0
    for path in (os.path.join(GOLD, "bam", "t001.bam"), os.path.join(GOLD, "bam", "t002.bam"), syn):
        raw, pos = open(path, "rb").read(), 0
        while pos < len(raw):
            bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
            comp, isize = raw[pos + 18:pos + bsize - 8], struct.unpack_from("<I", raw, pos + bsize - 4)[0]
            out = C.create_string_buffer(max(isize, 1))
            assert lib.tredbam_inflate_raw(comp, len(comp), out, isize) == 1
            assert out.raw[:isize] == (zlib.decompress(comp, -15) if isize else b"")
            pos += bsize
            n_blocks += 1
    assert n_blocks > 150


def test_bad_input_is_declined(lib):
    rng = random.Random(5)
    accepted = 0
    for _ in range(3000):                       # random bytes: declined, unless they happen to be a stream of that size
        data = os.urandom(rng.randrange(1, 3000))
        n = rng.choice([0, 5, 300, 4096, 65536])
        out = C.create_string_buffer(70000)
        if lib.tredbam_inflate_raw(data, len(data), out, n) == 1:
            d = zlib.decompressobj(-15)
            assert d.decompress(data) == out.raw[:n] and d.eof      # (an empty final block followed by junk, say)
            accepted += 1
    assert accepted < 100
    plain = os.urandom(3000) + b"hello world" * 500
    comp = zlib.compress(plain)[2:-4]
    for cut in range(len(comp)):                # truncated
        assert lib.tredbam_inflate_raw(comp[:cut], cut, C.create_string_buffer(9000), len(plain)) == 0
    for wrong in (len(plain) - 1, len(plain) + 1, 0):
        assert lib.tredbam_inflate_raw(comp, len(comp), C.create_string_buffer(9000), wrong) == 0
    ok = C.create_string_buffer(9000)
    assert lib.tredbam_inflate_raw(comp, len(comp), ok, len(plain)) == 1 and ok.raw[:len(plain)] == plain


def test_crc32_equals_zlibs(lib):
    """csrc/crc32_fold.h (carry-less-multiply folding; zlib's crc32 for tails and old CPUs) against zlib.crc32 at
    every length around the 16- and 64-byte lane boundaries, with and without a running value."""
    lib.tredbam_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_int64]
    lib.tredbam_crc32.restype = C.c_uint32
    rng = random.Random(11)
    for n in list(range(0, 260)) + [1000, 4095, 4096, 65535, 65536]:
        data = bytes(rng.randrange(256) for _ in range(n)) if n < 5000 else os.urandom(n)
        for init in (0, 0x9E3779B9):
            assert lib.tredbam_crc32(init, data, n) == zlib.crc32(data, init), (n, init)
    a, b = os.urandom(777), os.urandom(4242)
    assert lib.tredbam_crc32(lib.tredbam_crc32(0, a, len(a)), b, len(b)) == zlib.crc32(a + b)


def test_damaged_blocks_are_errors_not_genotypes(tmp_path):
    """A BGZF block is checked against its trailer after decoding (htslib does; pysam would raise): flipping a literal
    byte inside a stored block, or the CRC itself, fails the fetch; an ISIZE beyond 64 KiB is refused before anything
    is allocated."""
    def bgzf(payload, stored=True):
        comp = zlib.compressobj(0 if stored else 6, zlib.DEFLATED, -15)
        body = comp.compress(payload) + comp.flush()
        head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(body) + 25)
        return head + body + struct.pack("<II", zlib.crc32(payload), len(payload))
    src = open(os.path.join(GOLD, "bam", "t001.bam"), "rb").read()
    # re-block the file's first 60 kB of records as stored blocks so that a flipped byte stays a valid stream
    pos, plain = 0, b""
    while len(plain) < 60000:
        bsize = struct.unpack_from("<H", src, pos + 16)[0] + 1
        plain += zlib.decompress(src[pos + 18:pos + bsize - 8], -15)
        pos += bsize
    blocks = [bgzf(plain[i:i + 20000]) for i in range(0, len(plain), 20000)]
    eof = bgzf(b"", stored=False)

    def fetch_all(raw):
        path = str(tmp_path / "x.bam")
        with open(path, "wb") as fp:
            fp.write(raw)
        f = bamio.NativeAlignmentFile(path)
        try:
            return sum(1 for _ in f.fetch())
        finally:
            f.close()
    good = b"".join(blocks) + eof
    assert fetch_all(good) > 100
    bad = bytearray(good)
    bad[len(blocks[0]) + 18 + 5 + 9000] ^= 0x20            # a payload byte of the second (stored) block
    with pytest.raises(Exception, match="CRC"):
        fetch_all(bytes(bad))
    bad = bytearray(good)
    bad[len(blocks[0]) - 8] ^= 1                            # the first block's CRC field
    with pytest.raises(Exception, match="CRC"):
        fetch_all(bytes(bad))
    bad = bytearray(good)
    bad[len(blocks[0]) - 4:len(blocks[0])] = struct.pack("<I", 3000000000)   # ISIZE
    with pytest.raises(Exception, match="claims"):
        fetch_all(bytes(bad))
