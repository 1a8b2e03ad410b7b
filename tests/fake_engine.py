"""engine.Engine.genotype_packed without the kernels (CPU tests of the host side, tools/prof_host.py): result arrays of the
real shapes and densities -- a third of the reads tagged, marginals of 1-30 entries, ~80 joint entries per unit, and now
and then a unit without evidence (status 1) or one the grid rejects (status < 0) -- drawn from a seeded generator."""
import numpy as np

from tredparse_amd import _lib, engine as eng


class FakeEngine(object):
    def __init__(self, seed=1, odd_units=True):
        self.seed, self.odd_units, self.calls = seed, odd_units, 0

    def close(self):                                   # (engine.Engine.close: the CLI gives its context back before it ends)
        pass

    def genotype_packed(self, b, dense=False):
        rng = np.random.default_rng(self.seed + 1000 * self.calls)
        self.calls += 1
        r = eng.BatchResult()
        r.batch, r.grid, r.grid_off = b, None, None
        n, g = b.n_reads, b.n_units
        r.tag = np.where(rng.random(n) < 0.33, rng.integers(1, 6, n), 0).astype(np.uint8)
        r.h = rng.integers(1, 50, n).astype(np.int16)
        r.score = rng.integers(30, 150, n).astype(np.int16)
        hs = b.max_units + 2
        r.full, r.pref, r.rept = (np.zeros((g, hs), np.int32) for _ in range(3))
        r.calls = np.zeros(g, _lib.CALL_DTYPE)
        per = b.params["period"]
        h1 = rng.integers(5, 40, g)
        h2 = h1 + rng.integers(0, 60, g)
        r.calls["h1"], r.calls["h2"] = h1 * per, h2 * per
        r.calls["ci"] = np.stack([h1, h1 + 1, h2, h2 + 13], axis=1)
        r.calls["pp"] = np.where(rng.random(g) < 0.5, rng.random(g), rng.random(g) * 1e-7)
        r.calls["lik"], r.calls["n_pairs"] = -100.0 * rng.random(g), 521
        if self.odd_units:
            r.calls["status"][3::7] = 1
            r.calls["status"][5::11] = -3
            r.calls["pp"][2::13] = 1.0
        ms = 302
        r.marg = np.zeros((g, 2, ms), np.float64)
        r.marg[np.arange(g), 0, h1] = 1.0
        for u in range(g):
            w = int(rng.integers(1, 30))
            r.marg[u, 1, h2[u]:h2[u] + w] = rng.random(w) ** 4 + 1e-6
        cap = 80
        a = np.repeat(h1.astype(np.int64), cap)
        bb = (np.repeat(h2.astype(np.int64), cap) + np.tile(np.arange(cap, dtype=np.int64), g))
        v = rng.random(g * cap) / 40.0
        r.joint = [None] * g
        r.joint_units = (a, bb, v, np.arange(g, dtype=np.int64) * cap, rng.integers(0, cap + 1, g).astype(np.int32))
        return r

    def genotype_selected(self, scans, maxinsert=300, fullsearch=False, clip=False):
        """engine.Engine.genotype_selected over tests/walk_model.ModelInflater's selection: the scans' per-read arrays are
        filled from the model's picks (what tredgpu_genotype_selected brings back), the results are genotype_packed's."""
        b = eng._SelectedBatch()
        rows, counts = [], []
        for s in scans:
            dev, t0, sel = s.device
            lens, seqs, names = [], [], []
            for k in range(len(s.names)):
                a, q, nm = dev.inf.selected_reads(t0 + k)
                assert len(a) == sel["n_reads"][k]
                lens += a; seqs += q; names += nm
            s.read_len = np.array(lens, np.int32)
            s.seq4 = np.frombuffer(b"".join(seqs), np.uint8).copy()
            s.seq4_off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
            s.name_blob = b"".join(names)
            s.name_off = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
            s.name_id = np.zeros(len(lens), np.int32)
            row = np.zeros(len(s.names), _lib.UNIT_DTYPE)
            row["period"] = [len(x.repeat) for x in s.loci]
            rows.append(row)
            counts.append(sel["n_reads"])
        b.params = np.concatenate(rows)
        b.n_units = len(b.params)
        b.unit_read_off = np.zeros(b.n_units + 1, np.int32)
        np.cumsum(np.concatenate(counts), out=b.unit_read_off[1:])
        b.n_reads = int(b.unit_read_off[-1])
        b.clip, b.pair_id, b.ladder_keys = bool(clip), None, []
        b.max_units = max(-(-s.readlen // len(x.repeat)) for s in scans for x in s.loci)
        return self.genotype_packed(b)
