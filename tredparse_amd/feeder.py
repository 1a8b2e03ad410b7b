"""The GPU-inflate pipeline of a driver process: plan -> fill -> decode (+ record walks) on the device -> scans, three
inflaters deep (tred.run_many is its consumer; `tred.py --gpu-inflate [--gpu-walk]`).  Reference counterpart: pysam's
fetch / pileup under BamParser.parse and PEextractor (tredparse/bam_parser.py:184-257, 316-369), one sample x locus at a time."""
import atexit
import logging
import threading
import time
from concurrent.futures import Future, ThreadPoolExecutor

from .bam_parser import scan_sample
from .runtime import _options, collect_sample, mark, timing_add


# ---- scans over GPU-inflated blocks -----------------------------------------------------------------------------------
# Two thirds of a scan's host time is DEFLATE decoding of ~550 BGZF blocks per 30x sample, and the host's cores, not
# the GPU, bound the from-BAM rate.  With `inflate_device` set, run_many plans every sample's blocks from its index
# (bamio plan), has the GPU decode a whole chunk of samples in ONE launch (_lib.Inflater: one lane per block; kernels of
# different streams do not overlap on this GPU, so the batch is what fills it) and lets the scans take the blocks from
# the inflater's pinned output (bamio preload).  Blocks a plan misses, or the decoder rejects, are inflated by the scan
# itself as before: the results cannot differ.
def _plan_sample(arg, walk=False, select=False):
    """Thread: open the BAM and list the blocks its scan will read -- with walk, also the pair-length regions as tasks
    for the device's walk (bamio plan_walks / plan_blocks); with select, also what the device's read selection needs
    (_select_plan).  None: no GPU help for this sample."""
    from .bam_parser import DNAPE_ELONGATE, FLANKMATCH, SPAN, _site_arrays, open_bam, y_regions
    o = _options(arg)
    try:
        f = open_bam(o["bam"])
    except (IOError, ValueError):
        return None                                    # scan_sample reports the file
    try:
        if not hasattr(f, "plan"):
            raise ValueError("no native BAM layer")
        readlen = f.max_read_len(101)
        loci = [o["repo"][n] for n in o["names"]]
        sites, regions = _site_arrays(o["repo"], o["names"], loci, f)
        sexed = any(t.is_xlinked for t in loci)          # scan_sample then asks for the chrY depth windows too
        n, cbytes, obytes = f.plan(sites, regions, readlen, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN,
                                   use_alts=o["alts"] and not o["clip"], extra=y_regions(o["repo"].ref) if sexed else ())
        p = {"handle": f, "readlen": readlen, "n": n, "cbytes": cbytes, "obytes": obytes}
        if walk and n > 0:
            p["tasks"], p["chunks"] = f.plan_walks(sites, readlen, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
            p["alt_tasks"], p["alt_chunks"] = f.plan_alt_walks(sites, regions, readlen, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE,
                                                               span=SPAN, use_alts=o["alts"] and not o["clip"])
            p["coffset"], p["clen"], p["crc"], p["host"] = f.plan_blocks()
            p["select"] = _select_plan(o, f, loci, sites, regions, readlen, sexed, p) if select else None
        return p
    except Exception:
        f.close()
        return None


def _select_plan(o, f, loci, sites, regions, readlen, sexed, p):
    """What the read selection on the device (include/tredgpu.h section 5) needs of one sample beside its walk tables: a
    tredgpu_select_task per locus (the position range of bam_parser.py:209-213, the locus' alternative regions) and -- when
    a locus is X-linked -- one plain region task per chrY window of the sex inference, whose pile-up sums come back with
    the loci's.  None when this sample must go through the host's scan: options whose outputs need more than the device
    path returns (--log DEBUG prints every pair of the grid, --norepeatpairs needs the reads' name ids before the tally), a
    locus the file or the kernels cannot serve (its contig is missing, its template ladder or the reads are too long)."""
    import numpy as np
    from ._lib import SELECT_TASK_DTYPE
    from .bam_parser import MAX_READ_LEN, MAX_TEMPLATE_LEN, y_regions
    if o["log"] == "DEBUG" or not (o["repeatpairs"] or o["clip"]):
        return None
    if len(sites) == 0 or (sites["tid"] < 0).any() or (p["tasks"]["n_chunks"] < 0).any() or readlen > MAX_READ_LEN:
        return None
    if any(len(t.prefix) + t.period * -(-readlen // t.period) + len(t.suffix) > MAX_TEMPLATE_LEN for t in loci):
        return None
    use_alts = o["alts"] and not o["clip"]
    if use_alts and len(regions) and ((p["alt_tasks"]["n_chunks"] < 0) & (regions["tid"][:len(p["alt_tasks"])] >= 0)).any():
        return None                                    # (a region of a contig the file HAS that cannot be walked from the plan)
    sel = np.zeros(len(sites), SELECT_TASK_DTYPE)
    sel["pos_lo"] = np.maximum(sites["repeat_start"].astype(np.int64) - readlen, 0)
    sel["pos_hi"] = sites["repeat_end"].astype(np.int64) + readlen
    sel["alt_first"], sel["n_alt"] = sites["alt_first"], (sites["n_alt"] if use_alts else 0)
    out = {"sel": sel, "ytasks": None, "ychunks": None, "ywidth": None, "sexed": sexed}
    if sexed:
        ys = y_regions(o["repo"].ref)
        if all(f.tid(c) >= 0 for c, _, _ in ys):        # (a file without these contigs: the sex stays unknown, as in scan_sample)
            yt, yc = f.plan_region_walks(ys)
            if (yt["n_chunks"] < 0).any():
                return None
            out["ytasks"], out["ychunks"] = yt, yc
            out["ywidth"] = np.array([hi - lo + 1 for _, lo, hi in ys], np.float64)
    return out


def _scan_planned(arg, plan, out_addr, out_off, status, crc=None, pe=None, alt=None):
    """Thread: the sample's scan with its planned blocks preloaded from the inflater's output (crc: the decoder's
    checksums of those blocks -- the scan then does not walk the bytes for the BGZF CRC again; pe: the pair walks'
    results from the device, see scan_sample)."""
    o = _options(arg)
    f = plan["handle"]
    try:
        if status is not None:
            f.preload(out_addr, out_off, status, crc)
        return scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"], readlen=plan["readlen"], handle=f,
                           pe=pe, alt=alt)
    finally:
        if status is not None:
            hits, misses = f.preload_clear()
            timing_add(inflate_hits=hits, inflate_misses=misses)
        f.close()


# Inflaters are kept between run_many calls of a process (their pinned staging is ~45 MB per sample of a chunk, and
# page-locking it costs about a second per gigabyte): a feeder borrows three and gives them back.
_INFLATERS = {}
_INFLATERS_LOCK = threading.Lock()


def _borrow_inflaters(device, n, host_out=True):
    from ._lib import Inflater
    with _INFLATERS_LOCK:
        have = _INFLATERS.setdefault((device, host_out), [])
        out = [have.pop() for _ in range(min(n, len(have)))]
    while len(out) < n:
        out.append(Inflater(device, host_out=host_out))
    return out


def _return_inflaters(device, infs):
    with _INFLATERS_LOCK:
        for inf in infs:
            _INFLATERS.setdefault((device, getattr(inf, "host_out", True)), []).append(inf)


def pinned_bytes():
    """Page-locked host memory of the process's pooled inflaters (those a running feeder has borrowed are not counted)."""
    with _INFLATERS_LOCK:
        return sum(inf.pinned_bytes() for v in _INFLATERS.values() for inf in v if hasattr(inf, "pinned_bytes"))


def release_inflaters():
    """Frees the pooled inflaters (their pinned and device buffers)."""
    with _INFLATERS_LOCK:
        infs = [i for v in _INFLATERS.values() for i in v]
        _INFLATERS.clear()
    for inf in infs:
        inf.close()


atexit.register(release_inflaters)


class DeviceChunk(object):
    """The samples of one decode call whose reads were selected on the device: the inflater that holds them (its buffers must
    stay as they are until the genotyping call has packed the reads: `done()` gives it back to the feeder) and the call's two
    pair-length pools, which the samples' units index."""
    __slots__ = ("inf", "gp", "tp", "release")

    def __init__(self, inf, gp, tp):
        self.inf, self.gp, self.tp, self.release = inf, gp, tp, Future()

    def done(self):
        if not self.release.done():
            self.release.set_result(None)


def _device_scan(arg, p, dev, res, selres):
    """The SampleScan of a sample whose reads the device selected (what scan_sample returns, without the per-read arrays:
    engine.genotype_selected fills those in from the genotyping call): sex from the chrY regions' depth sums, per locus the
    depth, the read count and the slices of the call's pair-length pools."""
    import numpy as np
    from . import bamio
    from .bam_parser import SPAN, SampleScan
    o = _options(arg)
    sp = p["select"]
    s = SampleScan()
    s.path, s.names, s.loci = o["bam"], list(o["names"]), [o["repo"][n] for n in o["names"]]
    s.gender, s.ydepth, s.readlen, s.opened = "Unknown", -1, int(p["readlen"]), True
    if sp["sexed"] and sp["ytasks"] is not None:
        y = p["ytask_first"]
        s.ydepth = float(np.median(selres["depth_sum"][y:y + len(sp["ytasks"])] / sp["ywidth"]))
        s.gender = "Male" if s.ydepth > 1 else "Female"
    t, n = p["task_first"], len(p["tasks"])
    sel, r = selres[t:t + n], res[t:t + n]
    u = s.unit = np.zeros(n, bamio.SCAN_UNIT_DTYPE)
    u["n_reads"] = sel["n_reads"]
    u["read_first"] = np.cumsum(sel["n_reads"], dtype=np.int64) - sel["n_reads"]
    u["depth_sum"] = sel["depth_sum"]
    u["n_global"], u["n_target"] = r["n_global"], r["n_target"]
    u["global_first"], u["target_first"] = r["global_first"], r["target_first"]
    s.global_lens, s.target_lens = dev.gp, dev.tp
    window = np.array([x.repeat_end + SPAN - max(0, x.repeat_start - SPAN) + 1 for x in s.loci], np.float64)
    s.depth = u["depth_sum"] / window
    s.ploidy = np.array([1 if (s.gender == "Male" and x.is_xlinked) else x.ploidy for x in s.loci], np.int32)
    s.packed = s.word_off = s.read_len = s.seq4 = s.seq4_off = s.name_blob = s.name_off = s.name_id = None
    s.dropped = {}
    s.device = (dev, t, sel)
    return s


class _InflateFeeder(object):
    """Feeds run_many's chunks through plan -> GPU inflate -> scan, ahead of the consumer: next() returns the next
    (chunk, its scan futures), None behind the last one.  Three stages overlap: while the GPU decodes chunk k (a thread of its own makes the call,
    which sleeps through it), the feeder thread plans and fills chunk k + 1 into another inflater's staging, and the scan
    pool still reads chunk k - 1's blocks out of a third -- so there are three inflaters, each reused only when every
    scan that reads its output has finished.  close() can be called at any time -- also while the consumer is unwinding
    from an error: the threads are told to stop, whatever was planned but never handed to a scan is closed, and the
    inflaters go only after every scan that reads their buffers has ended."""
    SLOTS = 3

    def __init__(self, chunks, ex, device, walk=False, select=False):
        import queue
        self.chunks, self.ex, self.device, self.walk, self.select = chunks, ex, device, walk, bool(select and walk)
        self.on_device = []                            # DeviceChunks handed out and not yet released by the consumer
        # plans and fills have threads of their own: queued behind a chunk's 28 scans in the scan pool they started only
        # when those were done, and the pool then idled through the next chunk's decode
        self.prep = ThreadPoolExecutor(max_workers=2)
        self.gpu = ThreadPoolExecutor(max_workers=1)       # the decode calls, one after the other, in chunk order
        # (with the walks on the device only a fifth of the blocks come back: those inflaters keep no pinned copy of the whole
        #  output -- 45 MB per sample of a chunk -- and hand the wanted blocks over densely packed)
        self.inflaters = _borrow_inflaters(device, self.SLOTS, host_out=not walk)
        self.busy = [[] for _ in range(self.SLOTS)]
        self.decoding = [None] * self.SLOTS            # the slot's last decode job (it sets busy[slot] when it hands the scans out)
        self.q = queue.Queue(maxsize=2)
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._run, name="tred-inflate", daemon=True)
        self.thread.start()

    @staticmethod
    def _close_plans(plans):
        for p in plans:
            if p is not None:
                try:
                    p["handle"].close()
                except Exception:
                    pass

    def _prepare(self, ci, chunk):
        """Feeder thread: the chunk's plans, and their payloads in the staging of inflater ci % SLOTS."""
        slot = ci % self.SLOTS
        inf = self.inflaters[slot]
        if self.decoding[slot] is not None:
            self.decoding[slot].exception()            # chunk ci - SLOTS has been decoded and its scans are known ...
        for fut in self.busy[slot]:
            fut.exception()                            # ... and have ended (waits; the consumer sees the error itself)
        plans = [fut.result() for fut in [self.prep.submit(_plan_sample, a, self.walk, self.select) for a in chunk]]
        live = [p for p in plans if p is not None and p["n"] > 0]
        t0 = time.perf_counter()
        job = {"plans": plans, "live": live, "inf": inf, "slot": slot, "ooff": None, "n_all": 0}
        if live and not self.stop.is_set():
            try:
                n_all = sum(p["n"] for p in live)
                comp, out, coff, ooff = inf.reserve(sum(p["cbytes"] for p in live), sum(p["obytes"] for p in live), n_all)
                at = cb = ob = 0
                fills = []
                for p in live:
                    p["first"] = at
                    fills.append(self.prep.submit(p["handle"].plan_fill, inf.comp_addr, cb, ob, coff[at:at + p["n"] + 1],
                                                ooff[at:at + p["n"] + 1]))
                    at, cb, ob = at + p["n"], cb + p["cbytes"], ob + p["obytes"]
                for fut in fills:
                    fut.result()
                # (every sample wrote its own end as entry n: the next sample's first entry is the same number)
                job["ooff"], job["n_all"] = ooff, n_all
                if self.walk:
                    job["walk"] = self._walk_tables(live)
            except Exception as e:     # no GPU help for this chunk: the scans inflate for themselves
                logging.getLogger("tredparse_amd").warning("GPU inflate skipped for a chunk of %d samples (%s)", len(chunk), e)
        timing_add(inflate=time.perf_counter() - t0)
        return job

    @staticmethod
    def _walk_tables(live):
        """The chunk's pair-walk tasks: every sample's tables (bamio plan_walks / plan_blocks) moved to the sample's
        place among the call's blocks and chunks.  A sample with a select plan brings its chrY region tasks along, behind
        its loci's (`task_first` .. + len(tasks) are the loci, `ytask_first` the first region task), and the chunk then also
        has one tredgpu_select_task per task of the call."""
        import numpy as np
        from ._lib import SELECT_TASK_DTYPE

        def place(t, c, p, c0):
            t, c = t.copy(), c.copy()
            t["chunk_first"] += c0
            t["block_first"] += p["first"]
            t["block_end"] += p["first"]
            c["begin_block"][c["begin_block"] >= 0] += p["first"]
            return t, c
        tasks, chunks, sels, c0, t0 = [], [], [], 0, 0
        any_select = any(p.get("select") is not None for p in live)
        for p in live:
            t, c = place(p["tasks"], p["chunks"], p, c0)
            p["task_first"] = t0
            tasks.append(t)
            chunks.append(c)
            c0, t0 = c0 + len(c), t0 + len(t)
            sp = p.get("select")
            if any_select:
                sel = np.zeros(len(t), SELECT_TASK_DTYPE)
                sel["n_alt"] = -1                          # (a sample the host scans: its tasks are only walked for the pairs)
                sels.append(sel if sp is None else sp["sel"].copy())
            if sp is not None and sp["ytasks"] is not None:
                yt, yc = place(sp["ytasks"], sp["ychunks"], p, c0)
                p["ytask_first"] = t0
                tasks.append(yt)
                chunks.append(yc)
                ysel = np.zeros(len(yt), SELECT_TASK_DTYPE)
                ysel["n_alt"] = -1
                sels.append(ysel)
                c0, t0 = c0 + len(yc), t0 + len(yt)
        alt_tasks, alt_chunks, c0, t0 = [], [], 0, 0
        for p in live:
            t, c = place(p["alt_tasks"], p["alt_chunks"], p, c0)
            p["alt_first"] = t0
            alt_tasks.append(t)
            alt_chunks.append(c)
            c0, t0 = c0 + len(c), t0 + len(t)
        out = {"coffset": np.concatenate([p["coffset"] for p in live]), "clen": np.concatenate([p["clen"] for p in live]),
               "crc": np.concatenate([p["crc"] for p in live]), "tasks": np.concatenate(tasks), "chunks": np.concatenate(chunks),
               "alt_tasks": np.concatenate(alt_tasks), "alt_chunks": np.concatenate(alt_chunks), "select": None}
        if any_select:
            sel = np.concatenate(sels)
            # a locus' alternative regions are entries of the CALL's alt tasks: the sample's first one is added
            for p in live:
                if p.get("select") is not None:
                    a = p["task_first"]
                    sel["alt_first"][a:a + len(p["tasks"])] += p["alt_first"]
            out["select"] = sel
        return out

    def _decode_and_scan(self, chunk, job):
        """Decode thread: one launch for the chunk, then its scans go to the pool and their futures to the consumer."""
        plans, inf, handed, futs = job["plans"], job["inf"], 0, []
        try:
            if self.stop.is_set():
                return
            status = crc = walked = None
            out_addr, out_off = inf.out_addr, job["ooff"]
            if job["ooff"] is not None:
                t0 = time.perf_counter()
                mark("decode call", n=len(chunk))
                try:
                    if job.get("walk") is not None:
                        status, crc, walked, out_addr, out_off = self._run_walk(inf, job)
                    else:
                        status, crc = inf.run(job["n_all"], crc=True)
                        timing_add(inflate_blocks=job["n_all"], inflate_failed=int((status != 0).sum()))
                except Exception as e:
                    logging.getLogger("tredparse_amd").warning("GPU inflate skipped for a chunk of %d samples (%s)", len(chunk), e)
                    status = crc = None
                timing_add(inflate_gpu=time.perf_counter() - t0)
                mark("decoded", n=len(chunk))
            if self.stop.is_set():
                return
            dev = None
            for a, p in zip(chunk, plans):
                if p is None:
                    futs.append(self.ex.submit(collect_sample, a))
                elif status is None or p["n"] == 0:
                    futs.append(self.ex.submit(_scan_planned, a, p, 0, None, None))
                elif walked is not None and p.get("on_device"):
                    # its reads were selected where the blocks are: no scan -- a SampleScan without per-read arrays, which the
                    # genotyping call fills in (engine.genotype_selected), and the inflater stays this chunk's until then
                    if dev is None:
                        dev = DeviceChunk(inf, walked[1], walked[2])
                    done = Future()
                    done.set_result(_device_scan(a, p, dev, walked[0], walked[4]))
                    futs.append(done)
                    self._close_plans([p])
                else:
                    k = p["first"]
                    pe = alt = None
                    if walked is not None:
                        res, gp, tp, ares = walked[:4]
                        pe = (res[p["task_first"]:p["task_first"] + len(p["tasks"])], gp, tp)
                        alt = ares[p["alt_first"]:p["alt_first"] + len(p["alt_tasks"])]
                    futs.append(self.ex.submit(_scan_planned, a, p, out_addr, out_off[k:k + p["n"] + 1], status[k:k + p["n"]],
                                               crc[k:k + p["n"]], pe, alt))
                handed += 1                                # (that scan closes its own handle)
            self.busy[job["slot"]] = futs + ([dev.release] if dev is not None else [])
            if dev is not None:
                self.on_device = [d for d in self.on_device if not d.release.done()] + [dev]
            self._put((chunk, futs))
        except BaseException as e:     # hand the failure to the consumer instead of leaving it waiting
            self.busy[job["slot"]] = futs         # (scans already running read the slot's buffers: close() waits for them)
            self._put(e)
        finally:
            # only the plans no scan was given are closed here: a handle a running scan still uses must not be freed under it
            self._close_plans(plans[handed:])

    @staticmethod
    def _run_walk(inf, job):
        """Decode, walk the pair-length regions on the device -- and, with a selection, pick the loci's reads there --, fetch
        the blocks the scans still read (none for a sample whose selection went through: p["on_device"]).  Returns the
        statuses as the scans should see them (a block that was not fetched counts as not delivered), the checksums and
        the walk's (results, global pool, target pool, alternative loci's results, select results or None), and where the
        fetched blocks lie (address, offsets per block)."""
        import numpy as np
        from .bam_parser import walk_need
        w = job["walk"]
        from ._lib import walk_pool_pairs
        t0 = time.perf_counter()
        out = inf.run_walk(job["n_all"], w["coffset"], w["clen"], w["crc"], w["tasks"], w["chunks"], alt_tasks=w["alt_tasks"],
                           alt_chunks=w["alt_chunks"], pool_pairs=walk_pool_pairs(w["tasks"], job["ooff"]),
                           **({"select": w["select"]} if w.get("select") is not None else {}))
        status, crc, res, gp, tp, ares, alt_need = out[:7]
        selres = out[7] if len(out) > 7 else None
        full = int((res["status"] == 6).sum())
        if full:                           # (WALK_POOL_FULL: cannot happen with the bound above; a wrong plan would show here)
            logging.getLogger("tredparse_amd").warning("pair walk: %d of %d regions found the pair pool full and are walked on the host", full, len(res))
        t1 = time.perf_counter()
        need = np.zeros(job["n_all"], np.uint8)
        n_dev = 0
        for p in job["live"]:
            a, sp = p["first"], p.get("select")
            p["on_device"] = False
            if selres is not None and sp is not None:
                t = p["task_first"]
                ok = bool((selres["status"][t:t + len(p["tasks"])] == 0).all())
                if ok and sp["ytasks"] is not None:
                    y = p["ytask_first"]
                    ok = bool((selres["status"][y:y + len(sp["ytasks"])] == 0).all())
                p["on_device"] = ok
                n_dev += ok
            if not p["on_device"]:
                need[a:a + p["n"]] = walk_need(p["coffset"], p["host"], res[p["task_first"]:p["task_first"] + len(p["tasks"])],
                                               alt_need[a:a + p["n"]])
        t2 = time.perf_counter()
        out_addr, out_off = inf.out_addr, job["ooff"]
        if need.any() or n_dev == 0:
            if getattr(inf, "host_out", True):
                inf.fetch(need)
            else:
                out_addr, out_off = inf.fetch_dense(need)
        timing_add(walk_call=t1 - t0, walk_fetch=time.perf_counter() - t2)
        walkable = w["alt_tasks"]["n_chunks"] >= 0
        timing_add(walk_regions=len(res), walk_declined=int((res["status"] != 0).sum()), walk_blocks_fetched=int(need.sum()),
                   walk_alt_regions=int(walkable.sum()), walk_alt_declined=int((ares["status"][walkable] != 0).sum()),
                   inflate_blocks=job["n_all"], inflate_failed=int((status != 0).sum()), select_samples=n_dev,
                   select_declined=sum(1 for p in job["live"] if p.get("select") is not None) - n_dev if selres is not None else 0)
        return np.where(need != 0, status, 1).astype(np.int32), crc, (res, gp, tp, ares, selres), out_addr, out_off

    def _put(self, item):
        import queue
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        try:
            for ci, chunk in enumerate(self.chunks):
                if self.stop.is_set():
                    return
                job = self._prepare(ci, chunk)
                if self.stop.is_set():
                    self._close_plans(job["plans"])
                    return
                self.decoding[job["slot"]] = self.gpu.submit(self._decode_and_scan, chunk, job)
            self.gpu.submit(self._put, None)           # the end of the cohort, behind the last chunk's scans
        except BaseException as e:
            self._put(e)

    def next(self):
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def next_if_scanned(self):
        """The next chunk if it is waiting AND all of its scans have finished, else None (the end of the cohort and errors
        stay where they are, for next())."""
        with self.q.mutex:
            head = self.q.queue[0] if self.q.queue else None
            if not isinstance(head, tuple) or not all(f.done() for f in head[1]):
                return None
        return self.q.get()       # (one consumer: what was at the head still is)

    def close(self):
        import queue
        self.stop.set()
        while True:                                        # make room: a put in progress returns at once
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.thread.join()                                 # (bounded: the threads check the flag between every two steps)
        self.gpu.shutdown(wait=True)
        for dev in self.on_device:                         # (the consumer -- this thread -- is done with them, whatever it did)
            dev.done()
        for slot in self.busy:
            for fut in slot:
                fut.exception()                            # scans still reading the staging buffers: let them end
        self.prep.shutdown()
        _return_inflaters(self.device, self.inflaters)     # (kept for the process's next cohort; release_inflaters frees them)
        self.inflaters = []
