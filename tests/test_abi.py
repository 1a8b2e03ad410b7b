"""CPU: the C-ABI library loads, exports every symbol include/tredgpu.h declares, its structs have the
layout the binding assumes, and it fails loudly (no CPU fallback) without a GPU.  No compute calls."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from tredparse_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tredgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tredgpu_[a-z_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (tredgpu_[a-z_]+)", out))
    assert exported == set(names)


def test_bam_layer_header_symbols_all_exported():
    """include/tredbam.h (host-only BAM file layer, libtredbam.so): every declared entry point is exported."""
    from tredparse_amd import bamio
    src = open(os.path.join(ROOT, "include", "tredbam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tredbam_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 37 and "tredbam_plan_region_walks" in names and {"tredbam_emit_sample_files", "tredbam_emit_last_error", "tredbam_pairwise_sum", "tredbam_pe_pool_sizes"} <= set(names) and {"tredbam_plan_walks", "tredbam_plan_blocks", "tredbam_scan_pe", "tredbam_plan_alt_walks", "tredbam_scan_walked"} <= set(names) and {"tredbam_plan", "tredbam_plan_fill", "tredbam_preload", "tredbam_preload_crc", "tredbam_preload_clear"} <= set(names) and "tredbam_pair_stats" in names and "tredbam_sparse_json_many" in names and "tredbam_crc32" in names and "tredbam_details_json" in names and "tredbam_sparse_json" in names and "tredbam_scan" in names and "tredbam_max_read_len" in names and "tredbam_inflate_raw" in names
    out = subprocess.check_output(["nm", "-D", "--defined-only", bamio._LIB_PATH]).decode()
    assert set(re.findall(r" T (tredbam_[a-z0-9_]+)", out)) == set(names)
    assert ctypes.sizeof(ctypes.c_int32) * 10 + 4 == bamio._REC.size == 44      # tredbam_rec
    assert ctypes.sizeof(bamio.ScanOpts) == 32 and ctypes.sizeof(bamio.Pools) == 4 * 8 + 10 * 8   # scan structs
    # the native writer's structs (tredbam_emit_*): sizes as the C compiler lays them out
    assert ctypes.sizeof(bamio.EmitLocus) == 4 * 8 + 8 * 4 and ctypes.sizeof(bamio.EmitBatch) == 5 * 8 + 8 + 5 * 8 + 8
    assert ctypes.sizeof(bamio.EmitSample) == 3 * 8 + 8 + 8 + 11 * 8 and ctypes.sizeof(bamio.EmitOpts) == 4 * 8 + 16


def test_no_oracle_in_product_library():
    """The shipped library must not link or reference the oracle / reference build."""
    out = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode()
    assert "oracle" not in out and "ssw" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "tredparse_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt and "libref_driver" not in txt, f


def test_struct_layouts():
    assert ctypes.sizeof(_lib.UnitParams) == 64 and _lib.UNIT_DTYPE.itemsize == 64
    assert ctypes.sizeof(_lib.Call) == 56 and _lib.CALL_DTYPE.itemsize == 56
    assert ctypes.sizeof(_lib.SwParams) == 32
    assert _lib.UNIT_DTYPE.fields["half_depth"][1] == _lib.UnitParams.half_depth.offset == 56
    assert _lib.CALL_DTYPE.fields["lik"][1] == _lib.Call.lik.offset == 40


def test_version_and_errors_without_gpu():
    lib = _lib.load()
    assert b"gfx950" in lib.tredgpu_version()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.TredGpuError) as ei:
        _lib.Context(0)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_pack_reads_layout():
    """tredgpu_pack_reads (host-only helper) against a pure-python statement of the record layout."""
    reads = ["ACGTNNACGT" * 15, "a", "", "TTTTGGGGCCCCAAAAN" * 3, "acgtRYKM"]
    packed, woff, rlen = _lib.pack_reads(reads)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for r, s in enumerate(reads):
        L = len(s)
        assert rlen[r] == L
        nb, nm = (L + 15) // 16, (L + 31) // 32
        assert woff[r + 1] - woff[r] == nb + nm
        rec = packed[woff[r]:woff[r + 1]]
        for i, ch in enumerate(s.upper()):
            isn = ch not in code
            assert ((int(rec[nb + i // 32]) >> (i % 32)) & 1) == int(isn)
            if not isn:
                assert ((int(rec[i // 16]) >> (2 * (i % 16))) & 3) == code[ch]
    # the vectorised packer used by the synthetic generator produces the same records
    from tredparse_amd import synth
    codes = np.stack([synth.encode(("ACGTN" * 40)[k:k + 150]) for k in range(5)])
    p2, w2, l2 = _lib.pack_codes(codes)
    p1, w1, l1 = _lib.pack_reads([synth.decode(c) for c in codes])
    assert np.array_equal(p1[:w1[-1]], p2) and np.array_equal(w1, w2) and np.array_equal(l1, l2)
