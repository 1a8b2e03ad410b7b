"""Test infrastructure: a stand-in for _lib.Inflater on machines without a GPU -- zlib for the blocks and a plain Python
walk over the inflated bytes that follows the task / chunk tables the way csrc/walk.hip's pair_walk_kernel does
(bamread.cpp walk_region + PairTable: tredparse/bam_parser.py:316-369).  Only tests use it: the host-side plumbing of
run_many(gpu_walk=True) is exercised here, the kernel itself in test_pairwalk_gpu.py."""
import struct
import zlib

import numpy as np

from tredparse_amd import _lib

REF_OPS = (0, 2, 3, 7, 8)      # M D N = X


class ModelInflater(object):
    made = []

    def __init__(self, device=0, host_out=True):
        ModelInflater.made.append(self)
        self.comp_addr = self.out_addr = 0
        self.walks = self.fetched = 0
        self.host_out = host_out

    def reserve(self, cb, ob, n):
        self.bufs = (np.zeros(cb + 64, np.uint8), np.zeros(ob + 64, np.uint8), np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64))
        self.comp_addr, self.out_addr = self.bufs[0].ctypes.data, self.bufs[1].ctypes.data
        return self.bufs[0][:cb], self.bufs[1][:ob], self.bufs[2], self.bufs[3]

    def _inflate(self, n):
        comp, _, coff, ooff = self.bufs
        self.dev = np.zeros(int(ooff[n]) + 64, np.uint8)
        status, crc = np.zeros(n, np.int32), np.zeros(n, np.uint32)
        for k in range(n):
            try:
                data = zlib.decompressobj(-15).decompress(bytes(comp[coff[k]:coff[k + 1]]))
            except zlib.error:
                status[k] = -1
                continue
            if len(data) != ooff[k + 1] - ooff[k]:
                status[k] = -2
                continue
            self.dev[ooff[k]:ooff[k + 1]] = np.frombuffer(data, np.uint8)
            crc[k] = zlib.crc32(data)
        return status, crc

    def run(self, n, crc=False):
        status, sums = self._inflate(n)
        self.bufs[1][:len(self.dev) - 64] = self.dev[:-64]
        return (status, sums) if crc else status

    def run_walk(self, n, bcoff, bclen, xcrc, tasks, chunks, pairs_per_task=2048, alt_tasks=None, alt_chunks=None, pool_pairs=None,
                 select=None):
        self.walks += 1
        self.pool_pairs = pool_pairs
        status, crc = self._inflate(n)
        ooff = self.bufs[3]
        out = out_bytes = self.dev.tobytes()
        self.out_bytes = out_bytes
        res = np.zeros(len(tasks), _lib.WALK_RESULT_DTYPE)
        gp, tp = [], []
        ok = lambda k: status[k] == 0 and crc[k] == xcrc[k]     # noqa: E731
        picked = [None] * len(tasks)                            # per task: (places of the window's selected records, depth sum)
        for t, T in enumerate(tasks):
            r = _walk(T, chunks, out, ooff, bcoff, bclen, ok, sel=None if select is None else select[t])
            if isinstance(r, int):
                res["status"][t] = r
                continue
            g, tt, nwin, vbeg, vend = r[:5]
            res[t] = (0, len(g), len(tt), nwin, len(gp), len(tp), vbeg, vend)
            gp += g
            tp += tt
            picked[t] = r[5] if len(r) > 5 else None
        # (tred._run_walk sizes the pools from the planned bytes: the bound must hold for every call)
        assert pool_pairs is None or (len(gp) <= pool_pairs and len(tp) <= max(pool_pairs // 8, 16 * len(tasks)) + 1024), (len(gp), len(tp), pool_pairs)
        out = (status, crc, res, np.array(gp, np.int32), np.array(tp, np.int32))
        if alt_tasks is None:
            return out
        ares = np.zeros(len(alt_tasks), _lib.ALT_RESULT_DTYPE)
        need = np.zeros(n, np.uint8)
        for t, T in enumerate(alt_tasks):
            r = _walk(T, alt_chunks, out_bytes, ooff, bcoff, bclen, ok, alt=True)
            if isinstance(r, int):
                ares["status"][t] = r
                continue
            if len(r) > 6:
                ares["status"][t] = 6
                continue
            ares["n"][t] = len(r)
            for m, (at, kb, ka) in enumerate(r):
                ares["vbeg"][t][m] = at
                need[kb:ka + 1] = 1
        if select is None:
            return out + (ares, need)
        # the read selection (csrc/walk.hip select_kernel): the window's picks, then the alternative regions' hits region by region
        selres = np.zeros(len(tasks), _lib.SELECT_RESULT_DTYPE)
        self.selected = [None] * len(tasks)
        coff_to_block = {int(c): k for k, c in enumerate(bcoff)}
        for t, (T, S) in enumerate(zip(tasks, select)):
            if res["status"][t] != 0:
                selres["status"][t] = res["status"][t]
                continue
            places, depth = picked[t]
            places = list(places)
            st = 0
            for q in range(int(S["alt_first"]), int(S["alt_first"]) + max(int(S["n_alt"]), 0)):
                if alt_tasks[q]["n_chunks"] < 0:
                    continue
                if ares["status"][q] != 0:
                    st = int(ares["status"][q])
                    break
                for at in ares["vbeg"][q][:ares["n"][q]]:
                    places.append(int(ooff[coff_to_block[int(at) >> 16]]) + (int(at) & 0xFFFF))
            if st == 0 and len(places) > _lib.SELECT_CAP:
                st = 8
            lens = [struct.unpack_from("<i", out_bytes, a0 + 4 + 16)[0] for a0 in places]
            if st == 0 and lens and max(lens) > 480:
                st = 9
            if st:
                selres["status"][t] = st
                continue
            names = [max(out_bytes[a0 + 4 + 8] - 1, 0) for a0 in places]
            selres[t] = (0, len(places), sum((L + 15) // 16 + (L + 31) // 32 for L in lens), sum((L + 1) // 2 for L in lens), sum(names),
                         max(lens + [0]), depth)
            self.selected[t] = places
        return out + (ares, need, selres)

    def selected_reads(self, task):
        """(read_len, 4-bit sequences, names) of the reads the selection of the last run_walk picked for `task`, in order."""
        out, lens, seqs, names = self.out_bytes, [], [], []
        for a0 in self.selected[task]:
            r = a0 + 4
            l_name, n_cig, L = out[r + 8], struct.unpack_from("<H", out, r + 12)[0], struct.unpack_from("<i", out, r + 16)[0]
            at = r + 32 + l_name + 4 * n_cig
            lens.append(L)
            seqs.append(out[at:at + (L + 1) // 2])
            names.append(out[r + 32:r + 32 + max(l_name - 1, 0)])
        return lens, seqs, names

    def fetch(self, need):
        assert self.host_out
        ooff = self.bufs[3]
        for k in np.flatnonzero(np.asarray(need)):
            self.bufs[1][ooff[k]:ooff[k + 1]] = self.dev[ooff[k]:ooff[k + 1]]
        self.fetched += int(np.count_nonzero(need))
        return 1

    def fetch_dense(self, need):
        """The wanted blocks one after the other in a buffer of their own (tredgpu_inflater_fetch_dense; no gap merging:
        any superset of the wanted blocks is a valid answer)."""
        assert not self.host_out
        ooff, need = self.bufs[3], np.asarray(need)
        size = np.where(need != 0, np.diff(ooff[:len(need) + 1]), 0)
        off = np.zeros(len(need) + 1, np.int64)
        np.cumsum(size, out=off[1:])
        self.dense = np.zeros(int(off[-1]) + 64, np.uint8)
        for k in np.flatnonzero(need):
            self.dense[off[k]:off[k + 1]] = self.dev[ooff[k]:ooff[k + 1]]
        self.fetched += int(np.count_nonzero(need))
        return self.dense.ctypes.data, off

    def close(self):
        pass


def _walk(T, chunks, out, ooff, bcoff, bclen, ok, alt=False, sel=None):
    """The pair walk of one region -> (global lens, target lens, window records, vbeg, vend), or -- alt -- the records
    whose mate lies on contig tstart within [win_lo, win_hi] as [(virtual offset, first block, last block)]; an int: the
    status the kernel ends the task with."""
    if T["n_chunks"] < 0:
        return 1
    hits = []
    size_of = lambda k: int(ooff[k + 1] - ooff[k])              # noqa: E731

    def tell(k, upos):
        return (int(bcoff[k]) + int(bclen[k])) << 16 if upos >= size_of(k) else (int(bcoff[k]) << 16) | upos

    pairs, order = {}, []
    nwin = vbeg = vend = 0
    picks, depth = [], 0
    for c in range(int(T["chunk_first"]), int(T["chunk_first"]) + int(T["n_chunks"])):
        k, upos, cend = int(chunks["begin_block"][c]), int(chunks["begin_upos"][c]), int(chunks["end_voffset"][c])
        if k < T["block_first"] or k >= T["block_end"]:
            return 1
        if not ok(k):
            return 2
        while True:
            at = tell(k, upos)
            if at >= cend:
                break
            kb = k if upos < size_of(k) else k + 1
            pieces = []
            for n in (4, None):
                if n is None:
                    n = struct.unpack_from("<i", out, pieces[0])[0]
                    if n < 32:
                        return 3
                addr = -1
                while n > 0:
                    if upos >= size_of(k):
                        if k + 1 >= T["block_end"] or bcoff[k] + bclen[k] != bcoff[k + 1]:
                            return 1
                        k += 1
                        if not ok(k):
                            return 2
                        upos = 0
                    if addr < 0:
                        addr = int(ooff[k]) + upos
                    piece = min(n, size_of(k) - upos)
                    upos += piece
                    n -= piece
                pieces.append(addr)
            r = pieces[1]
            size = struct.unpack_from("<i", out, pieces[0])[0]
            tid, pos, l_name, _mq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", out, r)
            if tid != T["tid"] or pos >= T["end"]:
                if tid > T["tid"] or (tid == T["tid"] and pos >= T["end"]):
                    break
                continue
            if l_seq < 0 or 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 > size:
                return 3
            cig = struct.unpack_from("<{}I".format(n_cig), out, r + 32 + l_name) if n_cig else ()
            rend = -1
            if not (flag & 4) and n_cig:
                rend = pos + sum(c >> 4 for c in cig if (c & 15) in REF_OPS)
            e = pos + 1 if (rend < 0 or rend <= pos) else rend
            if not e > T["start"]:
                continue
            if alt:
                mtid, mpos = struct.unpack_from("<ii", out, r + 20)
                if mtid == T["tstart"] and T["win_lo"] <= mpos <= T["win_hi"]:
                    hits.append((at, kb, k))
                continue
            if pos < T["win_hi"] and e > T["win_lo"]:
                if nwin == 0:
                    vbeg = at
                nwin += 1
                vend = tell(k, upos)
                if sel is not None:                   # (select_kernel: the pile-up sum and the window's picks, in file order)
                    if not (flag & (0x4 | 0x100 | 0x200 | 0x400)) and rend >= 0:
                        depth += rend - pos
                    if sel["n_alt"] >= 0 and ((flag & 4) or sel["pos_lo"] <= pos <= sel["pos_hi"]):
                        picks.append(pieces[0])
            if T["span"] <= 0 or not (flag & 1) or (flag & 4) or (flag & 0x400):
                continue
            name = out[r + 32:r + 32 + max(l_name - 1, 0)]
            p = pairs.get(name)
            if p is None:
                p = pairs[name] = []
                order.append(name)
            if len(p) < 2:
                lead = trail = 0
                for c in cig:
                    if (c & 15) == 4:
                        lead += c >> 4
                    elif (c & 15) != 5:
                        break
                for c in reversed(cig):
                    if (c & 15) == 4:
                        trail += c >> 4
                    elif (c & 15) != 5:
                        break
                p.append((pos, rend, bool(flag & 0x10), lead, trail))
            else:
                p.append(None)
    if alt:
        return hits
    g, t = [], []
    for name in order:
        p = pairs[name]
        if len(p) < 2:
            continue
        a, b = p[0], p[1]
        if a[2] or not b[2]:
            continue
        if b[1] < 0:
            return 5
        tlen = (b[1] + b[4]) - (a[0] - a[3])
        if tlen >= T["span"]:
            continue
        (t if (a[0] < T["tstart"] and b[1] > T["tend"]) else g).append(tlen)
    return (g, t, nwin, vbeg, vend) if sel is None else (g, t, nwin, vbeg, vend, (picks, depth))
