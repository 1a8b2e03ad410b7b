"""CPU: profile hygiene.  Every rocprofv3 summary of the current round under profiles/ carries the library_version of the
build it was taken on (tools/pmc_to_json.py asks the library itself), and all of them name ONE build: the tree's -- the
hash tredparse_amd/csrc/Makefile bakes into libtredgpu.so from the kernel sources, recomputed here from the sources."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"


def tree_hash():
    csrc = os.path.join(ROOT, "tredparse_amd", "csrc")
    h = hashlib.sha256()
    for name in ("capi.hip", "sw_ladder.hip", "grid.hip", "inflate_decode.hip", "walk.hip", "inflater_api.hip", "tredgpu_internal.h",
                 "inflater_internal.h"):
        with open(os.path.join(csrc, name), "rb") as fp:
            h.update(fp.read())
    with open(os.path.join(ROOT, "include", "tredgpu.h"), "rb") as fp:
        h.update(fp.read())
    return h.hexdigest()[:16]


def test_makefile_hashes_the_files_this_test_hashes():
    mk = open(os.path.join(ROOT, "tredparse_amd", "csrc", "Makefile")).read()
    assert "SRCS := capi.hip sw_ladder.hip grid.hip inflate_decode.hip walk.hip inflater_api.hip" in mk
    assert "HDRS := tredgpu_internal.h inflater_internal.h ../../include/tredgpu.h" in mk
    assert "cat $(SRCS) $(HDRS) | sha256sum | cut -c1-16" in mk


def test_round_summaries_come_from_one_build_the_trees():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND + "_*pmc_summary.json")))
    if not paths:
        import pytest
        pytest.skip("no {} summaries under profiles/ yet".format(ROUND))
    versions = {}
    for p in paths:
        with open(p) as fp:
            versions[os.path.basename(p)] = json.load(fp).get("library_version", "")
    assert len(set(versions.values())) == 1, versions
    assert all(v.endswith("src " + tree_hash()) for v in versions.values()), (versions, tree_hash())


def test_round_campaign_records_name_the_trees_build_too():
    """the fuzz campaigns, the CLI rate and the rehearsal records of the round carry the library they ran on (their
    "library" field = tredgpu_version()) -- the same build as the counter summaries, the tree's."""
    import pytest
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND + "_fuzz_*.json")) + glob.glob(os.path.join(ROOT, "profiles", ROUND + "_cli_rate*.json")))
    if not paths:
        pytest.skip("no {} campaign records under profiles/ yet".format(ROUND))
    want = "src " + tree_hash()
    for p in paths:
        with open(p) as fp:
            rec = json.load(fp)
        lib = rec.get("library") or rec.get("library_version") or ""
        assert lib.endswith(want), (os.path.basename(p), lib, want)
