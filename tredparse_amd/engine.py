"""Batching engine: turns lists of sample x locus units into calls of the C ABI (libtredgpu.so).

The reference calls its kernels one alignment / one locus at a time (bam_parser.py:132-135,
tred.py:165-167); here the host mirrors (bam_parser.BamParser, models.IntegratedCaller, tred.run)
collect units and hand them to one Engine, which packs reads, registers the template ladders and
launches the SW, tally and grid kernels once per batch.  No CPU fallback: constructing an Engine
needs a GPU.
"""
import json
import os

import numpy as np

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL_VALUE = float(np.exp(-10))  # models.py:34


def load_model():
    """Step-size pdfs (6 x 37) and stutter weights (5) from the package data (models.py:42-84)."""
    with open(os.path.join(HERE, "data", "model.json")) as fp:
        m = json.load(fp)
    step = np.array([m["step_size_by_period"][str(p)] for p in range(1, 7)], np.float64)
    return step, np.array(m["stutter_weights"], np.float64)


class Unit(object):
    """Everything the hot path needs for one sample x locus (one runBam, tred.py:153-169)."""

    def __init__(self, tred, readlen, reads, depth, ploidy, global_lens, target_lens, maxinsert=300,
                 fullsearch=False, clip=False, read_pair_ids=None):
        self.tred = tred
        self.readlen = int(readlen)
        self.reads = list(reads)
        self.depth = float(depth)
        self.ploidy = int(ploidy)
        self.global_lens = list(global_lens)
        self.target_lens = list(target_lens)
        self.maxinsert = int(maxinsert)
        self.fullsearch = bool(fullsearch)
        self.clip = bool(clip)
        self.read_pair_ids = read_pair_ids  # only for --norepeatpairs
        period = len(tred.repeat)
        self.max_units = -(-self.readlen // period)          # bam_parser.py:73
        self.ref_len = tred.repeat_end - tred.repeat_start + 1  # bam_parser.py:68
        self.minpe = tred.repeat_end - tred.repeat_start + 2 * 9 + 2   # bam_parser.py:361


JOINT_CAP = 512   # sparse joint entries per unit asked for first (pairs with exp(ml - max) >= e^-10)


class PackedUnits(object):
    """A batch of sample x locus units already in the C ABI's layout (include/tredgpu.h): what
    bam_parser.scan_sample produces, concatenated over samples.  No read strings, no per-read Python objects.

      packed / read_off / read_len          the reads, unit after unit
      unit_read_off, ladder_keys[unit]      reads of each unit; its template ladder (prefix, repeat, suffix, max_units)
      params (UNIT_DTYPE)                   grid inputs; pe_off / tl_off index global_lens / target_lens
      pair_id                               per read, for --norepeatpairs (None otherwise)
    """

    @classmethod
    def from_scans(cls, picks, maxinsert=300, fullsearch=False, clip=False, repeatpairs=True):
        """picks: [(SampleScan, [locus indices])] -- the listed loci of each scan, in that order.  One pass of array
        operations per scan (a unit at a time in Python this cost a driver 0.6-1.2 ms per sample, 30 units each)."""
        b = cls()
        b.clip, b.ladder_keys = bool(clip), []
        words, offs, lens, ids, gls, tls, rows, counts = [], [np.zeros(1, np.int64)], [], [], [], [], [], []
        wbase = n_gl = n_tl = 0
        for scan, ks in picks:
            if not len(ks):
                continue
            ks = np.asarray(ks, np.int64)
            st, u = _static(scan), scan.unit[ks]
            first, n = u["read_first"].astype(np.int64), u["n_reads"].astype(np.int64)
            reads = _ranges(first, n)                         # pool indices of the units' reads, unit after unit
            if isinstance(reads, slice):
                w0, w1 = int(scan.word_off[reads.start]), int(scan.word_off[reads.stop])
                words.append(scan.packed[w0:w1])
                offs.append(scan.word_off[reads.start + 1:reads.stop + 1] - w0 + wbase)
                wbase += w1 - w0
            else:
                lo, hi = scan.word_off[reads], scan.word_off[reads + 1]
                size = hi - lo
                words.append(scan.packed[_ranges(lo, size, force=True)])
                offs.append(np.cumsum(size) + wbase)
                wbase += int(size.sum())
            lens.append(scan.read_len[reads])
            if not (repeatpairs or clip):
                ids.append(scan.name_id[reads])
            g, t = u["n_global"].astype(np.int64), u["n_target"].astype(np.int64)
            gls.append(scan.global_lens[_ranges(u["global_first"].astype(np.int64), g)])
            tls.append(scan.target_lens[_ranges(u["target_first"].astype(np.int64), t)])
            r = np.zeros(len(ks), _lib.UNIT_DTYPE)
            r["period"], r["readlen"], r["ploidy"] = st["period"][ks], scan.readlen, scan.ploidy[ks]
            r["maxinsert"], r["fullsearch"] = maxinsert, int(fullsearch)
            r["ref_len"], r["minpe"] = st["span"][ks] + 1, st["span"][ks] + 20
            r["cutoff_risk"], r["is_expansion"], r["is_recessive"] = st["cutoff_risk"][ks], st["is_expansion"][ks], st["is_recessive"][ks]
            r["pe_off"], r["n_global"] = n_gl + np.cumsum(g) - g, g
            r["tl_off"], r["n_target"] = n_tl + np.cumsum(t) - t, t
            r["half_depth"] = np.asarray(scan.depth, np.float64)[ks] / 2
            rows.append(r)
            counts.append(n)
            n_gl += int(g.sum())
            n_tl += int(t.sum())
            keys = st["ladder"].get(scan.readlen)
            if keys is None:
                keys = st["ladder"][scan.readlen] = [(x.prefix, x.repeat, x.suffix, -(-scan.readlen // len(x.repeat))) for x in scan.loci]
            b.ladder_keys += [keys[k] for k in ks.tolist()]
        cat = lambda parts, dt: np.ascontiguousarray(np.concatenate(parts), dt) if parts else np.zeros(0, dt)
        b.params = cat(rows, _lib.UNIT_DTYPE)
        b.n_units = len(b.params)
        b.unit_read_off = np.zeros(b.n_units + 1, np.int32)
        if counts:
            np.cumsum(np.concatenate(counts), out=b.unit_read_off[1:])
        b.n_reads = int(b.unit_read_off[-1])
        b.packed = cat(words, np.uint32)
        b.read_off = cat(offs, np.int64)
        b.read_len = cat(lens, np.int32)
        b.pair_id = None if (repeatpairs or clip) else cat(ids, np.int32)
        b.global_lens, b.target_lens = cat(gls, np.int32), cat(tls, np.int32)
        b.max_units = max([k[3] for k in b.ladder_keys] + [1])
        return b


def _ranges(first, count, force=False):
    """The concatenation of arange(first[i], first[i] + count[i]) -- as a slice when the ranges follow each other
    without a gap (the common case: a scan's pools hold its loci one after the other)."""
    total = int(count.sum())
    if total == 0:
        return slice(0, 0) if not force else np.zeros(0, np.int64)
    live = count > 0
    f, c = first[live], count[live]
    if not force and (len(f) == 1 or (f[1:] == f[:-1] + c[:-1]).all()):
        return slice(int(f[0]), int(f[0]) + total)
    starts = np.cumsum(c) - c
    return np.repeat(f - starts, c) + np.arange(total, dtype=np.int64)


def _static(scan):
    """Per-locus constants of a scan's locus list as arrays (period, tract span, cut-off, inheritance flags) and its
    template-ladder keys per read length -- the same for every sample of a cohort, so kept on the locus list's first
    Locus object (scans of one repo and list share their Locus objects)."""
    loci = scan.loci
    if not loci:
        return {"period": np.zeros(0, np.int32), "span": np.zeros(0, np.int32), "cutoff_risk": np.zeros(0, np.int32),
                "is_expansion": np.zeros(0, np.int32), "is_recessive": np.zeros(0, np.int32), "ladder": {}}
    key = tuple(id(t) for t in loci)
    hit = _STATIC.get(key)
    if hit is None or hit[0] is not loci[0]:
        st = {"period": np.array([len(t.repeat) for t in loci], np.int32),
              "span": np.array([t.repeat_end - t.repeat_start for t in loci], np.int32),
              "cutoff_risk": np.array([int(t.cutoff_risk) for t in loci], np.int32),
              "is_expansion": np.array([int(t.is_expansion) for t in loci], np.int32),
              "is_recessive": np.array([int(t.is_recessive) for t in loci], np.int32), "ladder": {}}
        if len(_STATIC) > 256:
            _STATIC.clear()
        hit = _STATIC[key] = (loci[0], st)
    return hit[1]


_STATIC = {}


class _SelectedBatch(object):
    """What a BatchResult's users read of its batch, for units whose reads were packed on the device (genotype_selected)."""
    __slots__ = ("unit_read_off", "n_units", "n_reads", "params", "clip", "ladder_keys", "pair_id", "max_units")


class BatchResult(object):
    """Arrays of one genotyped PackedUnits batch; unit(i) gives the per-unit view the callers format."""
    __slots__ = ("batch", "tag", "h", "score", "full", "pref", "rept", "calls", "marg", "joint", "grid", "grid_off",
                 "joint_units", "repeatpairs", "_emit")

    def unit(self, i):
        b = self.batch
        a, e = int(b.unit_read_off[i]), int(b.unit_read_off[i + 1])
        r = UnitResult()
        r.tags, r.hs, r.scores = self.tag[a:e], self.h[a:e], self.score[a:e]
        r.full = r.pref = r.rept_hist = None
        r.rept = int(self.rept[i].sum())
        r.call = self.calls[i]
        r.grid = None
        if getattr(self, "grid", None) is not None and self.calls[i]["status"] == 0:
            r.grid = self.grid[self.grid_off[i]:self.grid_off[i] + self.calls[i]["n_pairs"]]
        r.joint = self.joint[i] if self.calls[i]["status"] == 0 else None
        ju = getattr(self, "joint_units", None)
        r.joint_units = None
        if ju is not None and self.calls[i]["status"] == 0:
            a, b, v, lo, n = ju
            r.joint_units = (a[lo[i]:lo[i] + n[i]], b[lo[i]:lo[i] + n[i]], v[lo[i]:lo[i] + n[i]])
        r.P_h1, r.P_h2 = self.marg[i, 0], self.marg[i, 1]
        return r


class UnitResult(object):
    """grid: the dense dump {h1, h2, ml1..ml4} per pair (only when asked for); joint: (triples {h1, h2, exp(ml - max)}
    of the pairs >= e^-10, total over all distinct pairs) -- what P_h1h2 is printed from."""
    __slots__ = ("tags", "hs", "scores", "full", "pref", "rept_hist", "rept", "call", "grid", "joint", "P_h1", "P_h2",
                 "joint_units")


class Engine(object):
    def __init__(self, device_id=0, ctx=None):
        self.ctx = ctx or _lib.Context(device_id)
        step, w = load_model()
        self.ctx.set_model(step, w)       # gc=.68, score=1.0 (models.py:106)
        self._ladders = None

    def close(self):
        self.ctx.close()

    # ---- packed batches (the product path) ------------------------------------------------------------
    def _register(self, keys):
        """Ladder index of every key; the context's ladder table only ever grows, so indices stay valid."""
        known = self._ladders or []
        index = {k: i for i, k in enumerate(known)}
        grew = False
        for k in keys:
            if k not in index:
                index[k] = len(known)
                known = known + [k]
                grew = True
        if grew or self._ladders is None:
            self.ctx.set_ladders(known)
            self._ladders = known
        return np.asarray([index[k] for k in keys], np.int32)

    def classify_packed(self, b):
        """SW + tagging of a PackedUnits batch: (tag u8[], h i16[], score i16[]) per read."""
        n = b.n_reads
        tag, h, sc = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.int16), np.zeros(max(n, 1), np.int16)
        if n:
            lad = self._register(b.ladder_keys)
            self.ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, lad, b.n_units,
                                 _lib.default_sw_params(clip=b.clip), tag, h, sc, None, 0)
        return tag[:n], h[:n], sc[:n]

    def genotype_packed(self, b, dense=False):
        """The whole path for a PackedUnits batch -> BatchResult (sparse joint distribution included): one fused call
        (tredgpu_genotype_batch_joint: tags, histograms and calls stay on the device between the stages; one wait).
        dense: also the four terms of every (h1, h2) pair (a second grid call with the dump; what --log DEBUG prints) --
        that path goes through the three separate calls, which hand the histograms back."""
        if dense or b.n_units == 0:
            return self._genotype_packed_stepwise(b, dense)
        r = BatchResult()
        r.batch = b
        r.grid = r.grid_off = None
        g, n = b.n_units, b.n_reads
        hs = b.max_units + 2
        ms = max(int(b.params["maxinsert"].max()), hs) + 2
        r.tag, r.h, r.score = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.int16), np.zeros(max(n, 1), np.int16)
        r.full = r.pref = None
        r.rept = np.zeros((g, hs), np.int32)
        lad = self._register(b.ladder_keys)
        gl = b.global_lens if len(b.global_lens) else np.zeros(1, np.int32)
        tl = b.target_lens if len(b.target_lens) else np.zeros(1, np.int32)
        params = _lib.default_sw_params(clip=b.clip)
        r.calls = np.zeros(g, _lib.CALL_DTYPE)
        r.marg = np.zeros((g, 2, ms), np.float64)
        cap = np.full(g, JOINT_CAP, np.int64)
        while True:
            joff = np.zeros(g + 1, np.int64)
            joff[1:] = np.cumsum(cap)
            trip = np.zeros((int(joff[-1]), 3), np.float64)
            jn, jt = np.zeros(g, np.int32), np.zeros(g, np.float64)
            self.ctx.genotype_batch_joint(b.packed, b.read_off, b.read_len, n, b.unit_read_off, lad, b.params, g, params,
                                          b.pair_id if n else None, gl, len(b.global_lens), tl, len(b.target_lens),
                                          r.tag, r.h, r.score, hs, r.rept, r.calls, r.marg, ms, joff, trip, jn, jt)
            if (jn <= cap).all():
                break
            cap = np.maximum(cap, jn)         # a flat likelihood surface: ask again with room for every entry
        r.tag, r.h, r.score = r.tag[:n], r.h[:n], r.score[:n]
        r.joint = [(trip[joff[i]:joff[i] + jn[i]], float(jt[i])) for i in range(g)]
        per = np.repeat(b.params["period"].astype(np.int64), cap)
        with np.errstate(divide="ignore", invalid="ignore"):
            r.joint_units = (trip[:, 0].astype(np.int64) // per, trip[:, 1].astype(np.int64) // per, trip[:, 2] / np.repeat(jt, cap),
                             joff[:-1], jn)
        return r

    def genotype_selected(self, scans, maxinsert=300, fullsearch=False, clip=False):
        """genotype_packed for samples whose reads the device selected and still holds (feeder._device_scan: SampleScans with
        `.device` = (DeviceChunk, first task, select results)): one tredgpu_genotype_selected call -- pack on the device, SW +
        tagging -> histograms -> grid, one wait -- which also brings back the selected reads' lengths, 4-bit sequences and
        names; they are filled into the scans (per-sample views), so that the writers find what scan_sample would have
        left there.  Every locus of every scan is a unit, in order.  Returns the BatchResult (units in scan order)."""
        import ctypes as C
        r = BatchResult()
        r.grid = r.grid_off = r.full = r.pref = None
        segs, rows, keys, pools, sels = [], [], [], {}, []
        g_all, t_all, n_gl, n_tl = [], [], 0, 0
        for s in scans:
            dev, t0, sel = s.device
            n = len(s.names)
            if dev not in pools:
                pools[dev] = (n_gl, n_tl)
                g_all.append(dev.gp)
                t_all.append(dev.tp)
                n_gl, n_tl = n_gl + len(dev.gp), n_tl + len(dev.tp)
            gb, tb = pools[dev]
            tasks = np.arange(t0, t0 + n, dtype=np.int32)
            if segs and segs[-1][0] is dev.inf:
                segs[-1][1].append(tasks)
            else:
                segs.append((dev.inf, [tasks]))
            st, u = _static(s), s.unit
            row = np.zeros(n, _lib.UNIT_DTYPE)
            row["period"], row["readlen"], row["ploidy"] = st["period"], s.readlen, s.ploidy
            row["maxinsert"], row["fullsearch"] = maxinsert, int(fullsearch)
            row["ref_len"], row["minpe"] = st["span"] + 1, st["span"] + 20
            row["cutoff_risk"], row["is_expansion"], row["is_recessive"] = st["cutoff_risk"], st["is_expansion"], st["is_recessive"]
            row["pe_off"], row["n_global"] = gb + u["global_first"], u["n_global"]
            row["tl_off"], row["n_target"] = tb + u["target_first"], u["n_target"]
            row["half_depth"] = np.asarray(s.depth, np.float64) / 2
            rows.append(row)
            sels.append(sel)
            k = st["ladder"].get(s.readlen)
            if k is None:
                k = st["ladder"][s.readlen] = [(x.prefix, x.repeat, x.suffix, -(-s.readlen // len(x.repeat))) for x in s.loci]
            keys += k
        params = np.concatenate(rows)
        sel = np.concatenate(sels)
        g = len(params)
        uro = np.zeros(g + 1, np.int32)
        uwo, uso, uno = (np.zeros(g + 1, np.int64) for _ in range(3))
        np.cumsum(sel["n_reads"], out=uro[1:])
        np.cumsum(sel["n_words"], out=uwo[1:])
        np.cumsum(sel["seq4_bytes"], out=uso[1:])
        np.cumsum(sel["name_bytes"], out=uno[1:])
        n = int(uro[-1])
        b = r.batch = _SelectedBatch()
        b.unit_read_off, b.n_units, b.n_reads, b.params, b.clip, b.ladder_keys, b.pair_id = uro, g, n, params, bool(clip), keys, None
        b.max_units = max([k[3] for k in keys] + [1])
        hs = b.max_units + 2
        ms = max(int(params["maxinsert"].max()), hs) + 2
        r.tag, r.h, r.score = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.int16), np.zeros(max(n, 1), np.int16)
        r.rept = np.zeros((g, hs), np.int32)
        lad = self._register(keys)
        gl = np.ascontiguousarray(np.concatenate(g_all), np.int32) if n_gl else np.zeros(1, np.int32)
        tl = np.ascontiguousarray(np.concatenate(t_all), np.int32) if n_tl else np.zeros(1, np.int32)
        sw = _lib.default_sw_params(clip=clip, max_read_len=max(int(sel["max_len"].max()), 1))
        r.calls = np.zeros(g, _lib.CALL_DTYPE)
        r.marg = np.zeros((g, 2, ms), np.float64)
        read_len = np.zeros(max(n, 1), np.int32)
        s4off, nmoff = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        seq4, names = np.zeros(max(int(uso[-1]), 1), np.uint8), np.zeros(max(int(uno[-1]), 1), np.uint8)
        seg_arg = [(inf, np.concatenate(t)) for inf, t in segs]
        cap = np.full(g, JOINT_CAP, np.int64)
        while True:
            joff = np.zeros(g + 1, np.int64)
            joff[1:] = np.cumsum(cap)
            trip = np.zeros((int(joff[-1]), 3), np.float64)
            jn, jt = np.zeros(g, np.int32), np.zeros(g, np.float64)
            self.ctx.genotype_selected(seg_arg, uro, uwo, uso, uno, lad, params, g, sw, gl, n_gl, tl, n_tl, r.tag, r.h, r.score, hs,
                                       r.rept, r.calls, r.marg, ms, joff, trip, jn, jt, read_len, s4off, seq4, nmoff, names)
            if (jn <= cap).all():
                break
            cap = np.maximum(cap, jn)
        r.tag, r.h, r.score = r.tag[:n], r.h[:n], r.score[:n]
        r.joint = [(trip[joff[i]:joff[i] + jn[i]], float(jt[i])) for i in range(g)]
        per = np.repeat(params["period"].astype(np.int64), cap)
        with np.errstate(divide="ignore", invalid="ignore"):
            r.joint_units = (trip[:, 0].astype(np.int64) // per, trip[:, 1].astype(np.int64) // per, trip[:, 2] / np.repeat(jt, cap),
                             joff[:-1], jn)
        # the reads' arrays, a sample at a time, as scan_sample leaves them (offsets from the sample's first read)
        u0 = 0
        for s in scans:
            m = len(s.names)
            a, e = int(uro[u0]), int(uro[u0 + m])
            s.read_len = read_len[a:e]
            s.seq4_off = s4off[a:e + 1] - s4off[a]
            s.seq4 = seq4[int(s4off[a]):int(s4off[e])] if e > a else np.zeros(0, np.uint8)
            s.name_off = nmoff[a:e + 1] - nmoff[a]
            s.name_blob = names[int(nmoff[a]):int(nmoff[e])].tobytes() if e > a else b""
            s.name_id = np.zeros(e - a, np.int32)          # (read only under --norepeatpairs, which the device path does not take)
            u0 += m
        return r

    def _genotype_packed_stepwise(self, b, dense=False):
        """genotype_packed through the three separate calls (SW, tally, grid)."""
        r = BatchResult()
        r.batch = b
        r.grid = r.grid_off = None
        r.tag, r.h, r.score = self.classify_packed(b)
        g, n = b.n_units, b.n_reads
        hs = b.max_units + 2
        r.full, r.pref, r.rept = (np.zeros((g, hs), np.int32) for _ in range(3))
        self.ctx.tally(_lib.MEM_HOST, r.tag if n else np.zeros(1, np.uint8), r.h if n else np.zeros(1, np.int16), n,
                       b.unit_read_off, g, b.pair_id if n else None, hs, r.full, r.pref, r.rept)
        ms = max(int(b.params["maxinsert"].max()) if g else 0, hs) + 2
        r.calls, r.marg, r.joint, r.joint_units = self._grid_arrays(b.params, hs, r.full, r.pref, r.rept, b.global_lens,
                                                                     b.target_lens, ms, units_form=True)
        if dense and g:
            gl = b.global_lens if len(b.global_lens) else np.zeros(1, np.int32)
            tl = b.target_lens if len(b.target_lens) else np.zeros(1, np.int32)
            r.grid_off = np.zeros(g + 1, np.int64)
            r.grid_off[1:] = np.cumsum(np.maximum(r.calls["n_pairs"], 1))
            r.grid = np.zeros((int(r.grid_off[-1]), 6), np.float64)
            again = np.zeros(g, _lib.CALL_DTYPE)
            self.ctx.likelihood_grid(_lib.MEM_HOST, b.params, g, hs, r.full, r.pref, r.rept, gl, len(b.global_lens), tl,
                                     len(b.target_lens), again, r.grid_off, r.grid, None, 0)
        return r

    def _grid_arrays(self, up, hs, full, pref, rept, gl, tl, ms, units_form=False):
        """likelihood_grid_joint over array inputs; grows the joint capacity when a flat surface needs it.
        units_form: also the joint entries of the whole batch as P_h1h2 prints them -- alleles in repeat units, values
        divided by their unit's total -- computed in one pass over the batch's array (a unit's share is three slices;
        per unit the same arithmetic was thirty small numpy calls per sample)."""
        g = len(up)
        ngl, ntl = len(gl), len(tl)
        gl = gl if ngl else np.zeros(1, np.int32)
        tl = tl if ntl else np.zeros(1, np.int32)
        calls = np.zeros(g, _lib.CALL_DTYPE)
        marg = np.zeros((g, 2, ms), np.float64)
        cap = np.full(g, JOINT_CAP, np.int64)
        while True:
            joff = np.zeros(g + 1, np.int64)
            joff[1:] = np.cumsum(cap)
            trip = np.zeros((int(joff[-1]), 3), np.float64)
            jn, jt = np.zeros(g, np.int32), np.zeros(g, np.float64)
            self.ctx.likelihood_grid_joint(_lib.MEM_HOST, up, g, hs, full, pref, rept, gl, ngl, tl, ntl, calls, marg, ms,
                                           joff, trip, jn, jt)
            if (jn <= cap).all():
                break
            cap = np.maximum(cap, jn)
        joint = [(trip[joff[i]:joff[i] + jn[i]], float(jt[i])) for i in range(g)]
        if not units_form:
            return calls, marg, joint
        per = np.repeat(up["period"].astype(np.int64), cap)
        with np.errstate(divide="ignore", invalid="ignore"):
            ju = (trip[:, 0].astype(np.int64) // per, trip[:, 1].astype(np.int64) // per, trip[:, 2] / np.repeat(jt, cap),
                  joff[:-1], jn)
        return calls, marg, joint, ju

    # ---- (1) SW + tagging only -----------------------------------------------------------------------
    def classify(self, units, want_dump=False):
        """Per unit: (tags u8[], h i16[], score i16[]) [+ dump].  units: list of Unit."""
        ladders, lad_index, unit_ladder = [], {}, []
        for u in units:
            key = (u.tred.prefix, u.tred.repeat, u.tred.suffix, u.max_units)
            if key not in lad_index:
                lad_index[key] = len(ladders)
                ladders.append(key)
            unit_ladder.append(lad_index[key])
        if self._ladders != ladders:
            self.ctx.set_ladders(ladders)
            self._ladders = ladders
        reads = [r for u in units for r in u.reads]
        packed, woff, rlen = _lib.pack_reads(reads)
        n = len(reads)
        uro = np.zeros(len(units) + 1, np.int32)
        uro[1:] = np.cumsum([len(u.reads) for u in units])
        tag = np.zeros(max(n, 1), np.uint8)
        h = np.zeros(max(n, 1), np.int16)
        sc = np.zeros(max(n, 1), np.int16)
        nt = max([2 * l[3] for l in ladders] + [1])
        dump = np.zeros((max(n, 1), nt, 6), np.int16) if want_dump else None
        clips = set(u.clip for u in units)
        if len(clips) > 1:
            raise ValueError("all units of one batch must share the clip setting")
        params = _lib.default_sw_params(clip=clips.pop() if clips else False)
        if n:
            self.ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, uro, np.asarray(unit_ladder, np.int32),
                                 len(units), params, tag, h, sc, dump, nt if want_dump else 0)
        if (tag[:n] == _lib.TAG_INVALID).any():
            raise _lib.TredGpuError("a read exceeds TREDGPU_MAX_READ_LEN")
        return tag[:n], h[:n], sc[:n], uro, (dump[:n] if want_dump else None)

    # ---- the whole path ------------------------------------------------------------------------------
    def genotype(self, units, want_grid=True):
        """SW + tagging -> histograms -> likelihood grid for a batch of units; returns [UnitResult]."""
        if not units:
            return []
        tag, h, sc, uro, _ = self.classify(units)
        n, g = len(tag), len(units)
        hs = max(u.max_units for u in units) + 2
        full = np.zeros((g, hs), np.int32)
        pref = np.zeros((g, hs), np.int32)
        rept = np.zeros((g, hs), np.int32)
        pair_ids = None
        if any(u.read_pair_ids is not None for u in units):
            pair_ids = np.concatenate([np.asarray(u.read_pair_ids if u.read_pair_ids is not None
                                                  else -np.ones(len(u.reads)), np.int32) for u in units])
        self.ctx.tally(_lib.MEM_HOST, tag if n else np.zeros(1, np.uint8), h if n else np.zeros(1, np.int16), n, uro,
                       g, pair_ids, hs, full, pref, rept)
        calls, marg, dump, goff, joint = self._grid(units, hs, full, pref, rept, want_grid)
        out = []
        for i, u in enumerate(units):
            r = UnitResult()
            r.tags, r.hs, r.scores = tag[uro[i]:uro[i + 1]], h[uro[i]:uro[i + 1]], sc[uro[i]:uro[i + 1]]
            r.full = {int(k): int(v) for k, v in enumerate(full[i]) if v}
            r.pref = {int(k): int(v) for k, v in enumerate(pref[i]) if v}
            r.rept_hist = {int(k): int(v) for k, v in enumerate(rept[i]) if v}
            r.rept = int(rept[i].sum())
            r.call = calls[i]
            r.grid = dump[goff[i]:goff[i] + calls[i]["n_pairs"]] if dump is not None and calls[i]["status"] == 0 else None
            r.joint = joint[i] if joint is not None and calls[i]["status"] == 0 else None
            r.P_h1, r.P_h2 = marg[i, 0], marg[i, 1]
            out.append(r)
        return out

    def _grid(self, units, hs, full, pref, rept, want_grid, dense=False):
        """One grid call for the batch.  want_grid: also the joint distribution -- sparse (triples + total per
        unit, tredgpu_likelihood_grid_joint) unless dense=True asks for the full dump of every pair."""
        g = len(units)
        up = np.zeros(g, _lib.UNIT_DTYPE)
        gl, tl = [], []
        for i, u in enumerate(units):
            t = u.tred
            up[i] = (len(t.repeat), u.readlen, u.ploidy, u.maxinsert, int(u.fullsearch), u.ref_len, u.minpe,
                     int(t.cutoff_risk), int(t.is_expansion), int(t.is_recessive), len(gl), len(u.global_lens),
                     len(tl), len(u.target_lens), u.depth / 2)
            gl += u.global_lens
            tl += u.target_lens
        ngl, ntl = len(gl), len(tl)
        gl = np.asarray(gl or [0], np.int32)
        tl = np.asarray(tl or [0], np.int32)
        calls = np.zeros(g, _lib.CALL_DTYPE)
        ms = max(max(u.maxinsert for u in units), hs) + 2
        marg = np.zeros((g, 2, ms), np.float64)
        dump = goff = joint = None
        if want_grid and not dense:
            cap = np.full(g, JOINT_CAP, np.int64)
            while True:
                joff = np.zeros(g + 1, np.int64)
                joff[1:] = np.cumsum(cap)
                trip = np.zeros((int(joff[-1]), 3), np.float64)
                jn = np.zeros(g, np.int32)
                jt = np.zeros(g, np.float64)
                self.ctx.likelihood_grid_joint(_lib.MEM_HOST, up, g, hs, full, pref, rept, gl, ngl, tl, ntl, calls,
                                               marg, ms, joff, trip, jn, jt)
                if (jn <= cap).all():
                    break
                cap = np.maximum(cap, jn)     # a flat likelihood surface: ask again with room for every entry
            joint = [(trip[joff[i]:joff[i] + jn[i]], float(jt[i])) for i in range(g)]
        else:
            self.ctx.likelihood_grid(_lib.MEM_HOST, up, g, hs, full, pref, rept, gl, ngl, tl, ntl, calls, None, None,
                                     marg, ms)
        if want_grid and dense:
            goff = np.zeros(g + 1, np.int64)
            goff[1:] = np.cumsum(np.maximum(calls["n_pairs"], 1))
            dump = np.zeros((int(goff[-1]), 6), np.float64)
            calls2 = np.zeros(g, _lib.CALL_DTYPE)
            self.ctx.likelihood_grid(_lib.MEM_HOST, up, g, hs, full, pref, rept, gl, ngl, tl, ntl, calls2, goff, dump,
                                     None, 0)
        return calls, marg, dump, goff, joint

    def grid_from_counts(self, unit, full, pref, rept):
        """Likelihood grid of one unit from explicit histograms {units: count} (IntegratedCaller.call)."""
        keys = list(full) + list(pref) + [unit.max_units]
        hs = max(keys) + 2
        f = np.zeros((1, hs), np.int32)
        p = np.zeros((1, hs), np.int32)
        r = np.zeros((1, hs), np.int32)
        for k, v in full.items():
            f[0, k] = v
        for k, v in pref.items():
            p[0, k] = v
        r[0, 0] = rept
        calls, marg, dump, goff, _ = self._grid([unit], hs, f, p, r, True, dense=True)
        res = UnitResult()
        res.tags = res.hs = res.scores = None
        res.full, res.pref, res.rept_hist, res.rept = dict(full), dict(pref), {}, rept
        res.call = calls[0]
        res.grid = dump[goff[0]:goff[0] + calls[0]["n_pairs"]] if calls[0]["status"] == 0 else None
        res.joint = None
        res.P_h1, res.P_h2 = marg[0, 0], marg[0, 1]
        return res
