"""Model of the suffix-continuation formulation of the template-ladder Smith-Waterman (development checker).

A template prefix + repeat*u + suffix ends with the same |suffix| columns for every u.  Instead of sweeping
them once per u from the trunk state, one *reversed* alignment of the read against the reversed suffix gives,
for every read row, the best continuation of an alignment that leaves the trunk at that row (weight and end
cell, independent of u).  A template's result is then the maximum of
    the trunk's running best cell                     (alignment ends on the trunk),
    trunk H[i][c] + continuation by a match  WH[i]    (enters the suffix diagonally),
    trunk E[i][c+1] + continuation of a gap  WE[i]    (enters the suffix inside a horizontal gap),
    the best alignment lying entirely inside the suffix (shifted by c + 1 columns).
This script checks that formulation, in plain Python integers packed the way the kernel packs them, against the
CPU oracle (oracle/sw_oracle.c) on random and low-complexity inputs, several scorings.

    python tools/proto_continuation.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

K = 1 << 18
PAY = K - 1
NEG = -(1 << 40)


def sc(a, b, m, x):
    if a == "N" or b == "N":
        return 0
    return m if a == b else -x


def column(read, letter, col, H, E, m, x, go, ge, restricted):
    """One forward column (the kernel's recurrences, unscaled).  Returns H', E' (for col+1), ht."""
    L = len(read)
    ht = [0] * L
    for i in range(L):
        S = sc(read[i], letter, m, x) * K
        t1 = (H[i - 1] if i > 0 else NEG) + S
        t2 = S + (col << 9 | i)
        ht[i] = max(t1, t2, E[i])
    Hn = [0] * L
    F = NEG
    for i in range(L):
        Hn[i] = max(ht[i], F)
        F = max(F - ge * K, ht[i] - go * K)
    En = [max(E[i] - ge * K, (ht[i] if restricted else Hn[i]) - go * K) for i in range(L)]
    return Hn, En, ht


def better(key, start, bk, bs):
    return key > bk or (key == bk and start > bs)


def ladder(read, prefix, repeat, suffix, max_units, scoring, floor):
    m, x, go, ge = scoring
    L = len(read)
    B = len(suffix)
    trunk = prefix + repeat * max_units
    # ---- continuation vectors from the reversed alignment (unrestricted Gotoh) ----
    rr = read[::-1]
    Hr, Er = [NEG] * L, [NEG] * L
    for y in range(B - 1):
        Hr, Er, _ = column(rr, suffix[B - 1 - y], y, Hr, Er, m, x, go, ge, restricted=False)
    y = B - 1
    Dm = [0] * L
    for xx in range(L):
        S = sc(rr[xx], suffix[0], m, x) * K
        Dm[xx] = max((Hr[xx - 1] if xx > 0 else NEG) + S, S + (y << 9 | xx))
    WH = [Dm[L - 2 - i] if i + 1 < L else NEG for i in range(L)]
    WE = [Er[L - 2 - i] + go * K if i + 1 < L else NEG for i in range(L)]
    # ---- best alignment inside the suffix alone ----
    Hs, Es = [NEG] * L, [NEG] * L
    sk, ss = floor, 0
    for j in range(B):
        Hs, Es, _ = column(read, suffix[j], j, Hs, Es, m, x, go, ge, restricted=True)
        for i in range(L):
            key = (Hs[i] & ~PAY) | (511 - j) << 9 | (511 - i)
            if key > sk:
                sk, ss = key, Hs[i] & PAY
    # ---- trunk sweep ----
    H, E = [NEG] * L, [NEG] * L
    bk, bs = floor, 0
    out = []
    u = 1
    for c, letter in enumerate(trunk):
        H, E, _ = column(read, letter, c, H, E, m, x, go, ge, restricted=True)
        for i in range(L):
            key = (H[i] & ~PAY) | (511 - c) << 9 | (511 - i)
            if key > bk:
                bk, bs = key, H[i] & PAY
        if c == len(prefix) + len(repeat) * u - 1:
            k, s = bk, bs
            const = ((511 - c - B) << 9) + (512 - L)
            for i in range(L):
                a1 = (H[i] & ~PAY) + WH[i] + const
                if WH[i] > NEG // 2 and better(a1, H[i] & PAY, k, s):
                    k, s = a1, H[i] & PAY
                ef = max(E[i], H[i] - go * K)
                a2 = (ef & ~PAY) + WE[i] + const
                if WE[i] > NEG // 2 and better(a2, ef & PAY, k, s):
                    k, s = a2, ef & PAY
            if sk != floor:
                k3, s3 = sk - ((c + 1) << 9), ss + ((c + 1) << 9)
                if better(k3, s3, k, s):
                    k, s = k3, s3
            if k == floor or (k >> 18) <= 0:
                out.append((0, -1, -1, 0, 0))
            else:
                out.append((k >> 18, (s >> 9) & 511, 511 - ((k >> 9) & 511), s & 511, 511 - (k & 511)))
            u += 1
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    scorings = [(1, 5, 7, 2), (1, 5, 7, 2), (2, 1, 2, 1), (1, 1, 1, 1), (3, 2, 4, 1), (1, 3, 2, 2)]
    bad = 0
    pairs = 0
    for case in range(n_cases):
        scoring = scorings[case % len(scorings)]
        alpha = "ACGT" if rng.random() < 0.5 else "AC"
        def rs(n):
            return "".join(rng.choice(list(alpha), n))
        period = int(rng.integers(1, 6))
        repeat = rs(period)
        prefix, suffix = rs(int(rng.integers(0, 9))), rs(int(rng.integers(1, 9)))
        max_units = int(rng.integers(1, 8))
        # read: a template instance with noise, or junk
        u0 = int(rng.integers(0, max_units + 2))
        src = prefix + repeat * u0 + suffix
        read = list(src[int(rng.integers(0, 3)):])
        for _ in range(int(rng.integers(0, 5))):
            if not read:
                break
            p = int(rng.integers(0, len(read)))
            r = rng.random()
            if r < 0.35:
                read[p] = rng.choice(list(alpha + "N"))
            elif r < 0.7:
                del read[p:p + int(rng.integers(1, 4))]
            else:
                read[p:p] = list(rs(int(rng.integers(1, 4))))
        read = "".join(read)[:40]
        if len(read) < 2:
            continue
        got = ladder(read, prefix, repeat, suffix, max_units, scoring, PAY)
        refs = [prefix + repeat * u + suffix for u in range(1, max_units + 1)]
        want = po.sw_pairs([read], refs, [0] * len(refs), list(range(len(refs))), scoring=scoring)
        for u, (g, w) in enumerate(zip(got, want)):
            pairs += 1
            if tuple(int(v) for v in w) != tuple(g):
                bad += 1
                if bad <= 10:
                    print("MISMATCH", scoring, read, prefix, repeat, suffix, "u", u + 1, "model", g, "oracle", tuple(int(v) for v in w))
    print("cases", n_cases, "pairs", pairs, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
