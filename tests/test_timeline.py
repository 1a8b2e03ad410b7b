"""CPU: the drivers' tuning timeline (tredparse_amd/runtime.py mark / timeline_dump: TRED_TIMELINE=<directory>) -- off by default
and free, on it writes one JSON file per process that tools/cli_rate.py --timeline reads."""
import json
import os

from tredparse_amd import runtime


def test_marks_are_dropped_without_the_switch(tmp_path, monkeypatch):
    monkeypatch.setattr(runtime, "_TIMELINE", None)          # (what the module holds when TRED_TIMELINE is not set at import)
    monkeypatch.setenv("TRED_TIMELINE", str(tmp_path))
    runtime.mark("anything", n=3)
    runtime.timeline_dump()
    assert os.listdir(tmp_path) == []


def test_marks_are_written_per_process(tmp_path, monkeypatch):
    monkeypatch.setattr(runtime, "_TIMELINE", [])
    monkeypatch.setenv("TRED_TIMELINE", str(tmp_path))
    runtime.mark("scanned", n=36)
    runtime.mark("genotyped", n=36)
    runtime.timeline_dump()
    with open(os.path.join(str(tmp_path), "timeline_{}.json".format(os.getpid()))) as fp:
        ev = json.load(fp)
    assert [e[1] for e in ev] == ["scanned", "genotyped"] and ev[0][2] == {"n": 36} and ev[0][0] <= ev[1][0]
