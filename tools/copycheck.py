#!/usr/bin/env python3
"""Overlap of this repo's host Python with the reference's text (build container only: reads /root/reference).

For every tredparse_amd/*.py against every tredparse/*.py of the reference: share of normalised lines (stripped,
blank and comment-only lines dropped) that also occur in the reference file, and the longest run of consecutive
identical lines.  Target: < 15 % shared, no identical block of 6 or more lines."""
import difflib
import glob
import os
import sys

REF = os.environ.get("TRED_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(path):
    out = []
    for line in open(path, encoding="latin-1"):
        s = line.strip()
        if s and not s.startswith("#"):
            out.append(s)
    return out


def main():
    refs = {os.path.basename(p): norm(p) for p in glob.glob(os.path.join(REF, "tredparse", "*.py"))}
    bad = 0
    for mine in sorted(glob.glob(os.path.join(ROOT, "tredparse_amd", "*.py"))):
        a = norm(mine)
        if not a:
            continue
        worst = (0.0, 0, "")
        for name, b in refs.items():
            sb = set(b)
            shared = sum(1 for x in a if x in sb and len(x) > 8) / float(len(a))
            m = difflib.SequenceMatcher(None, a, b, autojunk=False)
            block = max((blk.size for blk in m.get_matching_blocks()), default=0)
            if (shared, block) > worst[:2]:
                worst = (shared, block, name)
        flag = worst[0] >= 0.15 or worst[1] >= 6
        bad += flag
        print("{:28s} {:4d} lines  worst vs {:16s} shared {:5.1f} %  longest block {:2d} {}".format(
            os.path.basename(mine), len(a), worst[2], 100 * worst[0], worst[1], "<-- over" if flag else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
