"""The GPU-inflate pipeline of a driver process: plan -> fill -> decode (+ record walks) on the device -> scans, three
inflaters deep (tred.run_many is its consumer; `tred.py --gpu-inflate [--gpu-walk]`).  Reference counterpart: pysam's
fetch / pileup under BamParser.parse and PEextractor (tredparse/bam_parser.py:184-257, 316-369), one sample x locus at a time."""
import atexit
import logging
import threading
import time
from concurrent.futures import ThreadPoolExecutor

from .bam_parser import scan_sample
from .runtime import _options, collect_sample, timing_add


# ---- scans over GPU-inflated blocks -----------------------------------------------------------------------------------
# Two thirds of a scan's host time is DEFLATE decoding of ~550 BGZF blocks per 30x sample, and the host's cores, not
# the GPU, bound the from-BAM rate.  With `inflate_device` set, run_many plans every sample's blocks from its index
# (bamio plan), has the GPU decode a whole chunk of samples in ONE launch (_lib.Inflater: one lane per block; kernels of
# different streams do not overlap on this GPU, so the batch is what fills it) and lets the scans take the blocks from
# the inflater's pinned output (bamio preload).  Blocks a plan misses, or the decoder rejects, are inflated by the scan
# itself as before: the results cannot differ.
def _plan_sample(arg, walk=False):
    """Thread: open the BAM and list the blocks its scan will read -- with walk, also the pair-length regions as tasks
    for the device's walk (bamio plan_walks / plan_blocks).  None: no GPU help for this sample."""
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
        return p
    except Exception:
        f.close()
        return None


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


class _InflateFeeder(object):
    """Feeds run_many's chunks through plan -> GPU inflate -> scan, ahead of the consumer: next() returns the next
    (chunk, its scan futures), None behind the last one.  Three stages overlap: while the GPU decodes chunk k (a thread of its own makes the call,
    which sleeps through it), the feeder thread plans and fills chunk k + 1 into another inflater's staging, and the scan
    pool still reads chunk k - 1's blocks out of a third -- so there are three inflaters, each reused only when every
    scan that reads its output has finished.  close() can be called at any time -- also while the consumer is unwinding
    from an error: the threads are told to stop, whatever was planned but never handed to a scan is closed, and the
    inflaters go only after every scan that reads their buffers has ended."""
    SLOTS = 3

    def __init__(self, chunks, ex, device, walk=False):
        import queue
        self.chunks, self.ex, self.device, self.walk = chunks, ex, device, walk
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
        plans = [fut.result() for fut in [self.prep.submit(_plan_sample, a, self.walk) for a in chunk]]
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
        place among the call's blocks and chunks."""
        import numpy as np
        def moved(key_t, key_c, first_key):
            tasks, chunks, c0, t0 = [], [], 0, 0
            for p in live:
                t, c = p[key_t].copy(), p[key_c].copy()
                t["chunk_first"] += c0
                t["block_first"] += p["first"]
                t["block_end"] += p["first"]
                c["begin_block"][c["begin_block"] >= 0] += p["first"]
                p[first_key] = t0
                tasks.append(t)
                chunks.append(c)
                c0, t0 = c0 + len(c), t0 + len(t)
            return np.concatenate(tasks), np.concatenate(chunks)
        tasks, chunks = moved("tasks", "chunks", "task_first")
        alt_tasks, alt_chunks = moved("alt_tasks", "alt_chunks", "alt_first")
        return {"coffset": np.concatenate([p["coffset"] for p in live]), "clen": np.concatenate([p["clen"] for p in live]),
                "crc": np.concatenate([p["crc"] for p in live]), "tasks": tasks, "chunks": chunks, "alt_tasks": alt_tasks,
                "alt_chunks": alt_chunks}

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
            if self.stop.is_set():
                return
            for a, p in zip(chunk, plans):
                if p is None:
                    futs.append(self.ex.submit(collect_sample, a))
                elif status is None or p["n"] == 0:
                    futs.append(self.ex.submit(_scan_planned, a, p, 0, None, None))
                else:
                    k = p["first"]
                    pe = alt = None
                    if walked is not None:
                        res, gp, tp, ares = walked
                        pe = (res[p["task_first"]:p["task_first"] + len(p["tasks"])], gp, tp)
                        alt = ares[p["alt_first"]:p["alt_first"] + len(p["alt_tasks"])]
                    futs.append(self.ex.submit(_scan_planned, a, p, out_addr, out_off[k:k + p["n"] + 1], status[k:k + p["n"]],
                                               crc[k:k + p["n"]], pe, alt))
                handed += 1                                # (that scan closes its own handle)
            self.busy[job["slot"]] = futs
            self._put((chunk, futs))
        except BaseException as e:     # hand the failure to the consumer instead of leaving it waiting
            self.busy[job["slot"]] = futs         # (scans already running read the slot's buffers: close() waits for them)
            self._put(e)
        finally:
            # only the plans no scan was given are closed here: a handle a running scan still uses must not be freed under it
            self._close_plans(plans[handed:])

    @staticmethod
    def _run_walk(inf, job):
        """Decode, walk the pair-length regions on the device, fetch the blocks the scans still read.  Returns the
        statuses as the scans should see them (a block that was not fetched counts as not delivered), the checksums and
        the walk's (results, global pool, target pool), and where the fetched blocks lie (address, offsets per block)."""
        import numpy as np
        from .bam_parser import walk_need
        w = job["walk"]
        from ._lib import walk_pool_pairs
        t0 = time.perf_counter()
        status, crc, res, gp, tp, ares, alt_need = inf.run_walk(job["n_all"], w["coffset"], w["clen"], w["crc"], w["tasks"], w["chunks"],
                                                                alt_tasks=w["alt_tasks"], alt_chunks=w["alt_chunks"],
                                                                pool_pairs=walk_pool_pairs(w["tasks"], job["ooff"]))
        full = int((res["status"] == 6).sum())
        if full:                           # (WALK_POOL_FULL: cannot happen with the bound above; a wrong plan would show here)
            logging.getLogger("tredparse_amd").warning("pair walk: %d of %d regions found the pair pool full and are walked on the host", full, len(res))
        t1 = time.perf_counter()
        need = np.zeros(job["n_all"], np.uint8)
        for p in job["live"]:
            a = p["first"]
            need[a:a + p["n"]] = walk_need(p["coffset"], p["host"], res[p["task_first"]:p["task_first"] + len(p["tasks"])],
                                           alt_need[a:a + p["n"]])
        t2 = time.perf_counter()
        if getattr(inf, "host_out", True):
            inf.fetch(need)
            out_addr, out_off = inf.out_addr, job["ooff"]
        else:
            out_addr, out_off = inf.fetch_dense(need)
        timing_add(walk_call=t1 - t0, walk_fetch=time.perf_counter() - t2)
        walkable = w["alt_tasks"]["n_chunks"] >= 0
        timing_add(walk_regions=len(res), walk_declined=int((res["status"] != 0).sum()), walk_blocks_fetched=int(need.sum()),
                   walk_alt_regions=int(walkable.sum()), walk_alt_declined=int((ares["status"][walkable] != 0).sum()),
                   inflate_blocks=job["n_all"], inflate_failed=int((status != 0).sum()))
        return np.where(need != 0, status, 1).astype(np.int32), crc, (res, gp, tp, ares), out_addr, out_off

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
        for slot in self.busy:
            for fut in slot:
                fut.exception()                            # scans still reading the staging buffers: let them end
        self.prep.shutdown()
        _return_inflaters(self.device, self.inflaters)     # (kept for the process's next cohort; release_inflaters frees them)
        self.inflaters = []
