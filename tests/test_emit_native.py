"""CPU: the native writer of a sample's files (libtredbam.so tredbam_emit_sample_files, tred.Emitter) against the Python
path it replaces (tred.format_scans + to_json + to_vcf, themselves pinned to the reference's outputs by
tests/test_e2e_gpu.py, test_vcf_golden.py, test_flags_gpu.py): same JSON bytes, same VCF text, for synthetic BAMs scanned
by the real front end and result arrays of the kernels' shapes (tests/fake_engine.py: units without evidence, units the
grid rejects, PP = 1.0 and 1e-8, empty and 30-entry marginals, 0..80 joint entries), with and without --norepeatpairs,
and for the samples the native printers hand back (a BAM that does not open)."""
import glob
import gzip
import os

import numpy as np
import pytest

from tests.fake_engine import FakeEngine
from tredparse_amd import bamio, synth_bam, tred
from tredparse_amd.meta import TREDsRepo

pytestmark = pytest.mark.skipif(bamio._native() is None, reason="libtredbam.so not built")


@pytest.fixture(scope="module")
def cohort(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("emit"))
    loci = [l for l in synth_bam.bench_loci() if l["name"] in ("HD", "DM1", "SCA10", "FXS", "ULD", "AR", "SBMA", "FRDA")]
    synth_bam.make_bams(root, 5, seed=11, loci=loci, workers=2)
    return root, [l["name"] for l in loci]


def _run(cohort, out, native, repeatpairs=True, extra=()):
    root, names = cohort
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    bams = sorted(glob.glob(os.path.join(root, "*.bam"))) + list(extra)
    tasks = [(os.path.basename(b)[:-4], b, repo, names, 300, False, False, True, repeatpairs, "ERROR") for b in bams]
    os.makedirs(out)
    cwd = os.getcwd()
    os.chdir(out)
    seen = []
    try:
        engine = FakeEngine(seed=5)
        if native:
            emit = tred.Emitter("hg38", repo, names, workers=2, on_sample=seen.append)
            try:
                tred.run_many(tasks, engine, batch=3, threads=2, lazy_details=True, emit=emit)
            finally:
                emit.close()
        else:
            tred.run_many(tasks, engine, batch=3, threads=2, lazy_details=True,
                          sink=lambda r: (tred.write_vcf_json(r, "hg38", repo, names, quiet=True), seen.append(r)))
    finally:
        os.chdir(cwd)
    return seen


def _files(d):
    out = {}
    for p in sorted(os.listdir(d)):
        with open(os.path.join(d, p), "rb") as fp:
            raw = fp.read()
        out[p] = gzip.decompress(raw) if p.endswith(".gz") else raw
    return out


@pytest.mark.parametrize("repeatpairs", [True, False])
def test_native_files_equal_the_python_path(cohort, tmp_path, repeatpairs, capsys):
    missing = str(tmp_path / "nothere.bam")
    py = _run(cohort, str(tmp_path / "py"), False, repeatpairs, extra=[missing])
    na = _run(cohort, str(tmp_path / "na"), True, repeatpairs, extra=[missing])
    a, b = _files(str(tmp_path / "py")), _files(str(tmp_path / "na"))
    assert sorted(a) == sorted(b) and len(a) == 10                 # five samples: JSON + VCF each; the missing BAM: nothing
    for name in a:
        assert a[name] == b[name], name
    # the native VCF file: ONE gzip member (RFC 1952) whose trailer carries the text's CRC-32 and length
    import struct, zlib
    for name in (n for n in b if n.endswith(".gz")):
        with open(os.path.join(str(tmp_path / "na"), name), "rb") as fp:
            raw = fp.read()
        assert raw[:4] == b"\x1f\x8b\x08\x00"
        d = zlib.decompressobj(-15)
        body = d.decompress(raw[10:])
        assert d.eof and len(d.unused_data) == 8 and body == b[name]
        assert struct.unpack("<II", d.unused_data) == (zlib.crc32(body), len(body))
    text = a["syn0000.json"].decode()
    assert '.P_h1h2": {' in text and '.details": [' in text and '"inferredGender"' in text and '.P_h1": ""' in text
    # what bench.py's check reads from the emitter: which loci were printed and the shorter allele, per sample
    by_key = {r["samplekey"]: r["tredCalls"] for r in py}
    assert sorted(s["samplekey"] for s in na) == sorted(by_key)
    for s in na:
        calls = by_key[s["samplekey"]]
        assert s["printed"] == [n + ".1" in calls for n in s["names"]]
        assert [f for f, p in zip(s["first_allele"], s["printed"]) if p] == [calls[n + ".1"] for n in s["names"] if n + ".1" in calls]
    capsys.readouterr()


def test_echo_prints_the_json_in_sample_order(cohort, tmp_path, capsys):
    root, names = cohort
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    bams = sorted(glob.glob(os.path.join(root, "*.bam")))
    tasks = [(os.path.basename(b)[:-4], b, repo, names, 300, False, False, True, True, "ERROR") for b in bams]
    os.chdir(tmp_path)
    emit = tred.Emitter("hg38", repo, names, echo=True, workers=4)
    try:
        tred.run_many(tasks, FakeEngine(seed=5), batch=2, threads=2, lazy_details=True, emit=emit)
    finally:
        emit.close()
    out = capsys.readouterr().out
    want = "".join(open(os.path.basename(b)[:-4] + ".json").read() for b in bams)
    assert out == want


def test_pairwise_sum_is_numpys():
    import ctypes as C
    lib = bamio._native()
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 130, 255, 256, 257, 300, 302, 303, 511, 640, 1000, 4099]:
        for _ in range(5):
            a = np.ascontiguousarray(rng.random(n) ** 8 * 10.0 ** rng.integers(-12, 3, n))
            got = lib.tredbam_pairwise_sum(a.ctypes.data if n else None, n)
            assert got == float(a.sum()), n


def test_json_strings_are_escaped_like_json_dumps(tmp_path, cohort):
    """A sample key and path outside ASCII: the native text still equals json.dumps' (ensure_ascii escapes)."""
    import json
    import shutil
    root, names = cohort
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    src = sorted(glob.glob(os.path.join(root, "*.bam")))[0]
    odd = str(tmp_path / "déjà \"vu\" \U0001F9EC.bam")
    shutil.copy(src, odd)
    shutil.copy(src + ".bai", odd + ".bai")
    task = ("kéy\t\U0001F9EC", odd, repo, names, 300, False, False, True, True, "ERROR")
    os.chdir(tmp_path)
    emit = tred.Emitter("hg38", repo, names)
    try:
        tred.run_many([task], FakeEngine(seed=5), batch=1, threads=1, lazy_details=True, emit=emit)
    finally:
        emit.close()
    with open(task[0] + ".json") as fp:
        text = fp.read()
    got = json.loads(text)
    assert got["bam"] == odd and got["samplekey"] == task[0]
    assert json.dumps(odd) in text and json.dumps(task[0]) in text and text.isascii()
