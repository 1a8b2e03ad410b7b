"""The SW kernel's suffix-continuation formulation, as the plain-integer model of
tools/proto_continuation.py, against the CPU oracle: random ladders, low-complexity alphabets, several scorings."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_continuation_model_matches_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proto_continuation.py"), "600", "11"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout
