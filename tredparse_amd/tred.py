#!/usr/bin/env python3
"""Genotype tandem-repeat disease loci from Illumina BAMs on AMD MI355X GPUs.

Command-line compatible with tredparse's tred.py (same flags, same per-sample `<key>.json` and `<key>.tred.vcf.gz`
outputs, same JSON keys and formatting) -- but a run is organised around GPU batches, not around one
sample x locus at a time:

    host threads      one native scan per BAM (bam_parser.scan_sample: sex, read length, and for every locus
                      depth + selected reads, already 2-bit packed, + pair lengths)
    GPU batch         the units of up to --batch-samples samples -> SW + tagging -> histograms -> (h1,h2) grid
                      (engine.Engine.genotype_packed, three libtredgpu.so calls)
    formatting        per sample: tredCalls dict -> JSON / VCF files
    --gpus N          N processes, one per GPU, each with its own share of the samples (no communication)

Reference counterparts: flags tred.py:64-113; per-sample driver run() :180-278; JSON :296-313; VCF :281-293,
:316-374; input modes read_csv :401-440; main :451-539.  Not offered: S3 upload (--output_path is accepted and
ignored) and the HLI-internal "@sample" lookup.
"""
import argparse
import gzip
import json
import logging
import os
import shutil
import sys
import threading
import time
from collections import deque
from itertools import islice
from concurrent.futures import ThreadPoolExecutor
from datetime import date, timedelta

from . import __version__, bamio
from .bam_parser import Details, scan_sample, tally
from .meta import BUILDS, TREDsRepo
from .models import GridError, SparseDist, format_call, pair_summaries

logging.basicConfig()
logger = logging.getLogger(__name__)
from .runtime import TIMING, _options, collect_sample, mark, timeline_dump, timing_add                      # noqa: F401  (shared with feeder.py / emit.py)
from .feeder import (_InflateFeeder, _plan_sample, _scan_planned, pinned_bytes,                # noqa: F401
                     release_inflaters)
from .emit import Emitter                                                                  # noqa: F401


# (ID, Number, Type, Description) of the VCF meta lines, in file order
_VCF_INFO = (("RPA", "1", "String", "Repeats per allele"), ("END", "1", "Integer", "End position of variant"),
             ("MOTIF", "1", "String", "Canonical repeat motif"), ("NS", "1", "Integer", "Number of samples with data"),
             ("REF", "1", "Integer", "Reference copy number"), ("CR", "1", "Integer", "Disease copy number cutoff"),
             ("IH", "1", "String", "Inheritance"), ("RL", "1", "Integer", "Reference STR track length in bp"),
             ("VT", "1", "String", "Variant type"))
_VCF_FORMAT = (("GT", "1", "String", "Genotype"), ("GA", "1", "String", "Genotype with absolute copy numbers"),
               ("FR", "1", "String", "Full spanning reads aligned to locus"),
               ("PR", "1", "String", "Partial reads aligned to locus"),
               ("RR", "1", "String", "Repeat-only reads aligned to locus"),
               ("DP", "1", "Integer", "Mean read depth around locus"), ("FDP", "1", "Integer", "Full spanning read depth"),
               ("PDP", "1", "Integer", "Partial read depth"), ("RDP", "1", "Integer", "Repeat read depth"),
               ("PEDP", "1", "Integer", "Paired-end read depth"), ("CI", "1", "String", "95% conf interval of estimates"),
               ("PP", "1", "Float", "Posterior probability of disease"), ("LABEL", "1", "String", "Risk assessment"))
_SAMPLE_KEYS = "GT:GB:FR:PR:RR:DP:FDP:PDP:RDP:PEDP:CI:PP:LABEL"
INFO = "".join('##{}=<ID={},Number={},Type={},Description="{}">\n'.format(kind, *row)
               for kind, rows in (("INFO", _VCF_INFO), ("FORMAT", _VCF_FORMAT)) for row in rows)


class _Parser(argparse.ArgumentParser):
    def error(self, message):          # a usage error shows the full help, like the reference's parser
        sys.stderr.write("error: {}\n\n".format(message))
        self.print_help()
        sys.exit(1)


def set_argparse():
    names = sorted(TREDsRepo().names)
    p = _Parser(prog="tred.py", description=__doc__.split("\n\n")[0],
                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("infile", nargs="?", help="a BAM, a text file listing BAMs, or a CSV of samplekey,bam[,locus]")
    p.add_argument("--ref", choices=BUILDS, default="hg38", help="genome build the BAMs are aligned to")
    p.add_argument("--tred", action="append", choices=names, default=None, help="locus to call (repeatable); all if omitted")
    p.add_argument("--haploid", action="append", help="contig to treat as haploid (repeatable)")
    p.add_argument("--useclippedreads", action="store_true", help="count clipped reads as evidence")
    p.add_argument("--noalts", action="store_true", help="skip the alternative loci where repeat reads get mismapped")
    p.add_argument("--norepeatpairs", action="store_true", help="discard pairs whose reads are both repeat-only")
    p.add_argument("--log", choices=("INFO", "DEBUG"), default="INFO", help="log level")
    p.add_argument("--version", action="version", version="%(prog)s " + __version__)
    p.add_argument("--toy", action="store_true", help=argparse.SUPPRESS)
    g = p.add_argument_group("Performance options")
    g.add_argument("--cpus", type=int, default=None,
                   help="host threads scanning BAMs, per driver process (default: the usable CPUs -- affinity mask and cgroup "
                        "quota -- shared among the drivers; the reference's default is cpu_count() workers)")
    g.add_argument("--gpus", type=int, default=1, help="GPUs to spread the samples over")
    g.add_argument("--drivers", default="auto",
                   help="driver processes per GPU, each with --cpus scan threads and its share of the samples (auto: from "
                        "the usable CPUs and --gpus by shard.driver_plan, never more than one per 32 samples)")
    g.add_argument("--gpu", type=int, default=0, help="device index when --gpus is 1")
    g.add_argument("--batch-samples", type=int, default=None,
                   help="samples per GPU batch (default: 64; 12 with --gpu-inflate, where a batch is also one decode + walk call; 36 "
                        "with --gpu-select)")
    g.add_argument("--gpu-inflate", action="store_true",
                   help="inflate the BAMs' BGZF blocks on the GPU, --batch-samples samples per launch (needs --cpus > 1; "
                        "pays when several driver processes share the host's cores: see DESIGN.md 4.4)")
    g.add_argument("--gpu-walk", action="store_true",
                   help="with --gpu-inflate: the pair-length walks (PEextractor's +-10 kb regions) run on the GPU over the "
                        "blocks it inflated; only the blocks of the loci's windows and alternative loci come back to the host")
    g.add_argument("--gpu-select", action="store_true",
                   help="with --gpu-inflate --gpu-walk: read selection, depth and 2-bit packing run on the GPU as well, over the "
                        "records the walks listed; no block comes back to the host and no host scan runs for the samples the "
                        "device can serve (the others are scanned as before)")
    g.add_argument("--maxinsert", type=int, default=300, help="largest allele considered, in repeat units")
    g.add_argument("--fullsearch", action="store_true", help="evaluate every allele pair up to --maxinsert")
    g = p.add_argument_group("I/O options")
    g.add_argument("--workdir", default=os.getcwd(), help="directory for the outputs")
    g.add_argument("--cleanup", action="store_true", help="remove --workdir when finished")
    g.add_argument("--checkexists", action="store_true", help="skip samples whose JSON is already there")
    g.add_argument("--no-output", action="store_true", help="compute but write nothing")
    g.add_argument("--quiet-json", action="store_true",
                   help="do not echo every sample's JSON on stdout (the reference prints each one, tred.py:311-312; at cohort scale "
                        "the echo of 300 KB per sample is a fifth of the command's time): the files are written all the same")
    g = p.add_argument_group("AWS and Docker options")
    g.add_argument("--sample_id", help="sample id (names the outputs together with the execution id)")
    g.add_argument("--workflow_execution_id", help="workflow execution id")
    g.add_argument("--input_bam_path", help="input path; takes precedence over infile")
    g.add_argument("--output_path", help="accepted for compatibility, ignored (no S3 upload)")
    p.add_argument("--task-file", help=argparse.SUPPRESS)     # set by the --gpus parent for its children
    return p


# ---- inputs -----------------------------------------------------------------------------------------------------
_REMOTE = ("s3://", "http://", "https://", "ftp://")


def bam_path(bam):
    return bam if bam.startswith(_REMOTE) else os.path.abspath(bam)


def _stem(path):
    return os.path.basename(path).rsplit(".", 1)[0]


def read_csv(infile, args):
    """[(samplekey, bam, locus or None)] from any of the input forms:
    a .bam/.cram path (key = file stem, or executionid_sampleid when both ids are given); a text file with one BAM
    per line; a CSV with samplekey,bam and an optional third column naming the one locus to call for that row
    (rows whose second field is not a .bam path -- headers -- are skipped)."""
    if infile.startswith("@"):
        raise SystemExit("`@sample` inputs need HLI's internal sample database, which this build does not have")
    if infile.endswith((".bam", ".cram")):
        bam = bam_path(infile)
        ids = (args.workflow_execution_id, args.sample_id)
        return [("_".join(ids) if all(ids) else _stem(bam), bam, None)]
    with open(infile) as fp:
        rows = [line.strip() for line in fp.read().splitlines()]
    if rows and rows[0].endswith(".bam") and "," not in rows[0]:
        return [(_stem(r), bam_path(r), None) for r in rows]
    out = []
    for r in rows:
        cells = r.split(",")
        if len(cells) >= 2 and cells[1].endswith(".bam"):
            out.append((cells[0], bam_path(cells[1]), cells[2] if len(cells) == 3 else None))
    return out


def counter_s(counts):
    """{15: 4, 6: 1} -> '6|1;15|4'."""
    return ";".join("{}|{}".format(k, int(counts[k])) for k in sorted(counts))


# ---- one sample: scan -> units -> calls ---------------------------------------------------------------------------
def _skeleton(o, scan):
    calls = {"inferredGender": scan.gender, "depthY": scan.ydepth}
    if scan.opened:
        calls["readLen"] = scan.readlen
    return {"samplekey": o["samplekey"], "bam": o["bam"], "tredCalls": calls}


def _fill_unit(calls, scan, k, res, repeatpairs, pairs, lazy=False):
    """The 21 keys of one locus from its kernel results (pairs = models.pair_summaries(scan))."""
    t = scan.loci[k]
    call = format_call(t, res, lazy=lazy)            # may raise GridError: the locus is then left out
    counts, details, rept = tally(scan, k, res.tags, res.hs, repeatpairs=repeatpairs, lazy=lazy)
    n = t.name
    calls[n + ".1"], calls[n + ".2"] = call["alleles"]
    calls[n + ".FR"], calls[n + ".PR"] = counter_s(counts["FULL"]), counter_s(counts["PREF"])
    calls[n + ".RR"] = counter_s(counts["REPT"])
    calls[n + ".DP"] = float(scan.depth[k])
    calls[n + ".FDP"], calls[n + ".PDP"] = sum(counts["FULL"].values()), sum(counts["PREF"].values())
    calls[n + ".RDP"] = rept
    for key, v in pairs[k].items():
        calls[n + "." + key] = v
    for key in ("CI", "PP", "label", "P_h1", "P_h2", "P_h1h2"):
        calls[n + "." + key] = call[key]
    calls[n + ".details"] = details


def _py2_str(x):
    """str(float) as Python 2 prints it (12 significant digits): the reference's DEBUG lines are built with str()."""
    s = "%.12g" % x
    return s if any(c in s for c in ".en") else s + ".0"       # ('nan' and 'inf' have an n)


def debug_lines(scan, k, res):
    """--log DEBUG: the reference's only diagnostics -- one line per tagged read, `TAG: h=  n, seq=...`
    (bam_parser.py:177-178, logger BamParser; HANG reads included) and one per allele pair of the grid,
    `*** (h1, h2) ml1 ml2 ml3 ml4 ml` in repeat units (models.py:270-272, logger IntegratedCaller); a term the
    reference does not evaluate (no spanning reads, no partial reads, paired-end term off) prints as its literal 0."""
    from ._lib import TAG_NAMES
    rlog, glog = logging.getLogger("BamParser"), logging.getLogger("IntegratedCaller")
    rlog.setLevel(logging.DEBUG)
    glog.setLevel(logging.DEBUG)
    a, _ = scan.reads_of(k)
    for i, (t, h) in enumerate(zip(res.tags, res.hs)):
        name = TAG_NAMES.get(int(t))
        if name is not None:
            rlog.debug("%s: h=%3d, seq=%s", name, int(h), scan.sequence(a + i))
    if res.grid is None:
        return
    period = len(scan.loci[k].repeat)
    has_full = any(int(t) == 1 for t in res.tags)
    has_part = any(int(t) in (2, 3) for t in res.tags)
    run_pe = bool(res.call["run_pe"])
    for h1, h2, m1, m2, m3, m4 in res.grid:
        glog.debug(" ".join(["***", str((int(h1) // period, int(h2) // period)), _py2_str(m1) if has_full else "0",
                             _py2_str(m2) if has_part else "0", _py2_str(m3), _py2_str(m4) if run_pe else "0",
                             _py2_str(((m1 + m2) + m3) + m4)]))


def _genotype(engine, picks, o):
    """PackedUnits of `picks` through the GPU.  A batch that fails as a whole is retried sample by sample and then
    unit by unit, so that one unit the kernels reject costs only itself (the reference loses only the failing
    locus too).  Returns {scan index: [(BatchResult, first unit of the sample in it, its locus indices)]} -- one piece
    per sample unless the retries went down to single units."""
    from ._lib import TredGpuError
    from .engine import PackedUnits
    kw = dict(maxinsert=o["maxinsert"], fullsearch=o["fullsearch"], clip=o["clip"],
              repeatpairs=o["repeatpairs"] or o["clip"])
    out = {}

    def attempt(sub):
        t0 = time.perf_counter()
        batch = PackedUnits.from_scans([(s, ks) for _, s, ks in sub], **kw)
        timing_add(pack=time.perf_counter() - t0)
        if batch.n_units == 0:
            return
        br = engine.genotype_packed(batch, dense=True) if o["log"] == "DEBUG" else engine.genotype_packed(batch)
        br.repeatpairs = kw["repeatpairs"]
        i = 0
        for si, _, ks in sub:
            if ks:
                out.setdefault(si, []).append((br, i, list(ks)))
            i += len(ks)

    try:
        attempt(picks)
    except TredGpuError as e:
        logger.error("GPU batch failed (%s); retrying in smaller pieces", e)
        for si, s, ks in picks:
            try:
                attempt([(si, s, ks)])
            except TredGpuError:
                for k in ks:
                    try:
                        attempt([(si, s, [k])])
                    except TredGpuError as e1:
                        logger.error("Exception on `%s` %s (%s)", s.path, s.names[k], e1)
    return out


def genotype_scans(engine, task_args, scans):
    from ._lib import TredGpuError
    return _genotype_scans(engine, task_args, scans, TredGpuError)


def _genotype_scans(engine, task_args, scans, TredGpuError):
    """GPU half of a batch: the kernels' results for every unit of the scans, (picks, parts) with parts as _genotype
    returns them (unit_results turns them into per-unit views).
    The kernel-side options of a GPU batch are the batch's: tasks that differ in them go in separate batches (the CLI's
    are uniform; API callers of run_many may mix them)."""
    picks = [(si, s, [k for k in range(len(s.names)) if k not in s.dropped] if s.opened else [])
             for si, s in enumerate(scans)]
    t0 = time.perf_counter()
    groups = {}
    for pick, arg in zip(picks, task_args):
        o = _options(arg)
        # (a scan whose reads the device selected and still holds -- feeder._device_scan -- goes to the call that packs them there)
        on_device = getattr(pick[1], "device", None) is not None
        key = (o["maxinsert"], o["fullsearch"], o["clip"], o["repeatpairs"] or o["clip"], o["log"] == "DEBUG", on_device)
        groups.setdefault(key, (o, []))[1].append(pick)
    parts = {}

    def on_the_host(o, sub):
        """The samples of `sub` scanned by the host after all and genotyped the host-packed way (scans[] is the caller's list:
        the writers find the host's scan there)."""
        again = []
        for si, s, _ in sub:
            h = collect_sample(task_args[si])
            scans[si] = h
            picks[si] = (si, h, [k for k in range(len(h.names)) if k not in h.dropped] if h.opened else [])
            again.append(picks[si])
            s.device[0].done()
        parts.update(_genotype(engine, again, o))
    try:
        for key, (o, sub) in groups.items():
            if not key[-1]:
                parts.update(_genotype(engine, sub, o))
                continue
            try:
                parts.update(_genotype_selected(engine, sub, o))
            except TredGpuError as e:
                # the call over the device-held reads failed as a whole: those samples are scanned on the host after all and
                # go the host-packed way, whose retries cost a bad unit only itself
                logger.error("GPU batch over device-selected reads failed (%s); scanning its %d samples on the host", e, len(sub))
                on_the_host(o, sub)
                continue
            # A selected record without a sequence (SEQ '*': pysam gives None and the reference's len(seq) raises,
            # bam_parser.py:129-133) drops its locus in the host's scan (TREDBAM_UNIT_NO_SEQ); the device's selection does not
            # look for it, but the lengths it brings back show it: such a sample is the host's.
            odd = [p for p in sub if len(p[1].read_len) and int(p[1].read_len.min()) == 0]
            if odd:
                on_the_host(o, odd)
    finally:
        for _, s, _ in picks:                  # the inflaters that held the device's selections are the feeder's again
            dev = getattr(s, "device", None)
            if dev is not None:
                dev[0].done()
    timing_add(gpu=time.perf_counter() - t0, gpu_calls=len(groups))
    return picks, parts


def _genotype_selected(engine, picks, o):
    """_genotype for scans whose reads are on the device: one engine.genotype_selected call for all of them."""
    t0 = time.perf_counter()
    scans = [s for _, s, _ in picks]
    br = engine.genotype_selected(scans, maxinsert=o["maxinsert"], fullsearch=o["fullsearch"], clip=o["clip"])
    br.repeatpairs = True
    timing_add(pack=time.perf_counter() - t0)
    out, i = {}, 0
    for si, s, ks in picks:
        out[si] = [(br, i, list(ks))]
        i += len(s.names)
    return out


def unit_results(parts, only=None):
    """{(scan index, k): UnitResult} of genotype_scans' parts (only: that scan index alone)."""
    res = {}
    for si, pieces in parts.items():
        if only is not None and si != only:
            continue
        for br, i0, ks in pieces:
            for j, k in enumerate(ks):
                res[(si, k)] = br.unit(i0 + j)
    return res


def format_scans(task_args, scans, picks, res, lazy_details=False):
    """Host half of a batch: the result dicts in task order from the kernels' results."""
    t1 = time.perf_counter()
    results = []
    for si, (arg, scan) in enumerate(zip(task_args, scans)):
        o = _options(arg)
        result = _skeleton(o, scan)
        pairs = pair_summaries(scan) if picks[si][2] else None
        for k in picks[si][2]:
            if (si, k) not in res:
                continue
            if o["log"] == "DEBUG":
                debug_lines(scan, k, res[(si, k)])
            try:
                _fill_unit(result["tredCalls"], scan, k, res[(si, k)], o["repeatpairs"] or o["clip"], pairs,
                           lazy=lazy_details)
            except GridError as e:
                logger.error("Exception on `%s` %s (%s)", o["bam"], scan.names[k], e)
        results.append(result)
    timing_add(format=time.perf_counter() - t1)
    return results


def finish_batch(engine, task_args, scans, lazy_details=False):
    """GPU half + formatting for the scans of one batch; returns the result dicts in task order.  lazy_details:
    `<locus>.details` as bam_parser.Details views (list-like) and the sparse distributions as models.SparseDist
    (dict-like) instead of lists and dicts; to_json prints both natively."""
    if not task_args:
        return []
    picks, parts = genotype_scans(engine, task_args, scans)
    return format_scans(task_args, scans, picks, unit_results(parts), lazy_details=lazy_details)


def run(arg, engine=None):
    """One sample, same argument tuple and return value as the reference's run():
    (samplekey, bam, repo, locus names, maxinsert, fullsearch, clip, alts, repeatpairs, log) ->
    {'samplekey', 'bam', 'tredCalls'}."""
    from .engine import Engine
    return finish_batch(engine or Engine(), [arg], [collect_sample(arg)])[0]


class _Writer(object):
    """sink(result) calls moved off the driver thread: worker threads take them from a queue (JSON text, gzip and the
    file writes spend most of their time outside the interpreter lock, so they overlap the driver's next GPU batch
    and formatting); at most `depth` results wait.  One worker keeps the order of the calls; with more (sinks whose
    calls are independent, like one set of files per sample) a driver is no longer held to one thread's 3-5 ms per
    sample.  An exception in the sink is re-raised by close()."""

    def __init__(self, sink, depth=64, workers=1):
        import queue
        self.sink, self.q, self.error = sink, queue.Queue(maxsize=depth), None
        self.threads = [threading.Thread(target=self._run, name="tred-writer", daemon=True) for _ in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.error is None:
                t0 = time.perf_counter()
                try:
                    self.sink(item)
                except BaseException as e:      # handed to the driver thread
                    self.error = e
                timing_add(write=time.perf_counter() - t0)

    def __call__(self, result):
        self.q.put(result)

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.error is not None:
            raise self.error


MERGED_BATCH = 48        # samples per genotyping call at most when decoded chunks pile up (run_many)


def _chunked(task_args, first, batch):
    """Lists of `first`, then `batch` tasks, taken lazily from any iterable."""
    it = iter(task_args)
    size = first
    while True:
        chunk = list(islice(it, size))
        if not chunk:
            return
        yield chunk
        size = batch


def run_many(task_args, engine, pool=None, batch=64, sink=None, threads=1, lazy_details=False, ahead_batches=2,
             background_sink=False, inflate_device=None, gpu_walk=False, emit=None, gpu_select=False):
    """run() over many samples, `batch` samples per GPU batch.  task_args: a list, or any iterable of run() argument
    tuples (taken lazily, a chunk at a time: a cohort need not be known in advance).  BAMs are scanned by `threads` host
    threads (or the executor given as `pool`), up to `ahead_batches` batches ahead of the GPU (with a single batch in
    flight the scan threads idle whenever a batch does not divide evenly among them, and while the driver formats).  Each
    finished result goes to sink(result), or into the returned list.  lazy_details: see finish_batch.
    background_sink: sink runs on a writer thread (in order) instead of the driver thread; an integer > 1: on that many
    writer threads, in any order (the sink's calls must then be independent of each other).
    emit: an Emitter -- the samples' output files are then written natively from the batch's arrays and no result dict
    is built (sink is not called, nothing is returned).
    inflate_device: GPU that inflates the samples' BGZF blocks, a chunk of samples per launch (None: the scans inflate on
    the host); needs scan threads.  gpu_walk: the pair-length walks (PEextractor) run on that GPU too, over the blocks it
    just inflated; only the blocks of the loci's windows and of the alternative loci come back.  gpu_select (with gpu_walk):
    the read selection, the depth and the 2-bit packing happen there as well (include/tredgpu.h section 5) -- no block comes
    back and no host scan runs for a sample the device could serve; any other is scanned as before.
    (Measured and removed: the GPU half of a batch on a thread of its own beside the formatting of the previous batch --
    the two halves fight over the interpreter lock, 20.1-20.4 k against 20.5 k genotypes/s with five drivers, 23.2 k against
    27.5 k with six --, and several decode chunks per genotyping batch, 14.7 / 11.0 k against 25 k: docs/history.)"""
    n_known = len(task_args) if hasattr(task_args, "__len__") else None
    # (with the blocks inflated on a GPU the scan pool is wanted even for one thread: the feeder hands it the samples the
    #  device could not serve -- with gpu_select that is all a scan thread still does)
    own = pool is None and (threads > 1 or inflate_device is not None) and (n_known is None or n_known > 1)
    # the first GPU batch is only as large as one round of the scan threads: nothing else can start before it is in
    # (only when there is more than one batch anyway: an extra GPU call costs more than it hides on small inputs)
    first = min(batch, max(1, threads)) if (own or pool is not None) and (n_known is None or n_known > batch) else batch
    chunks = _chunked(task_args, first, batch)
    ex = ThreadPoolExecutor(max_workers=threads) if own else pool
    out = []
    writer = _Writer(sink, workers=int(background_sink)) if (background_sink and sink is not None) else None
    if writer is not None:
        sink = writer
    feeder = None
    if inflate_device is not None and ex is not None:
        try:
            feeder = _InflateFeeder(chunks, ex, inflate_device, walk=gpu_walk, select=gpu_select)
        except Exception as e:       # no pinned memory, no device ...: the scans inflate for themselves
            logging.getLogger("tredparse_amd").warning("GPU inflate not available (%s): BGZF blocks are inflated on the host", e)

    def scanned():
        """(chunk, its scans) in task order, the scans running ahead of the consumer."""
        if feeder is not None:
            while True:
                item = feeder.next()
                if item is None:
                    return
                chunk, futs = item
                t0 = time.perf_counter()
                scans = [f.result() for f in futs]
                timing_add(scan_wait=time.perf_counter() - t0)
                # Chunks whose scans are already in ride along in the same genotyping call.  A call's time is mostly per
                # CALL once the decoder's wavefronts fill the device (a dozen launches that each wait for room: 21 ms for
                # 12 samples against 21 ms for 1 000 on an idle device), so a driver that has fallen behind its decoder
                # catches up by taking two or three chunks at once -- and never waits for one to do so.
                while len(chunk) + batch <= max(batch, MERGED_BATCH):
                    more = feeder.next_if_scanned()
                    if more is None:
                        break
                    chunk = chunk + more[0]
                    scans = scans + [f.result() for f in more[1]]
                    timing_add(merged_chunks=1)
                yield chunk, scans
        elif ex is not None:
            ahead = deque()
            more = True
            while True:
                while more and len(ahead) <= max(1, ahead_batches):
                    chunk = next(chunks, None)
                    if chunk is None:
                        more = False
                    else:
                        ahead.append((chunk, [ex.submit(collect_sample, a) for a in chunk]))
                if not ahead:
                    return
                chunk, futs = ahead.popleft()
                t0 = time.perf_counter()
                scans = [f.result() for f in futs]
                timing_add(scan_wait=time.perf_counter() - t0)
                yield chunk, scans
        else:
            for chunk in chunks:
                yield chunk, [collect_sample(a) for a in chunk]

    def shut_down():
        if feeder is not None:
            feeder.close()
        if own:
            ex.shutdown()
    try:
        for chunk, scans in scanned():
            if emit is not None:
                mark("scanned", n=len(chunk))
                picks, parts = genotype_scans(engine, chunk, scans)
                mark("genotyped", n=len(chunk))
                t0 = time.perf_counter()
                for si, (arg, scan) in enumerate(zip(chunk, scans)):
                    emit.submit(arg, scan, parts.get(si, []))
                timing_add(format=time.perf_counter() - t0)
                mark("submitted", n=len(chunk))
                if emit.error is not None:        # a writer thread failed: stop scanning and genotyping the rest of the cohort
                    raise emit.error
                continue
            for r in finish_batch(engine, chunk, scans, lazy_details=lazy_details):
                if sink is not None:
                    sink(r)
                else:
                    out.append(r)
    except BaseException:
        # unwinding from an error: stop the helpers, keep THIS exception (a sink that also failed must not replace it)
        shut_down()
        if writer is not None:
            try:
                writer.close()
            except BaseException:
                pass
        raise
    shut_down()
    if writer is not None:
        writer.close()
    return out


# ---- outputs ------------------------------------------------------------------------------------------------------
def _flat(d, depth):
    """A dict whose values are all scalars, as json.dumps(sort_keys=True, indent=4, separators=(',', ': ')) prints
    it at nesting `depth` -- through the C encoder (indent=None keeps it in C; the item separator carries the line
    break and the indentation)."""
    if not d:
        return "{}"
    pad = " " * (4 * (depth + 1))
    body = json.dumps(d, sort_keys=True, separators=(",\n" + pad, ": "))
    return "{\n" + pad + body[1:-1] + "\n" + " " * (4 * depth) + "}"


_P8, _P12, _P16 = " " * 8, " " * 12, " " * 16


def _flat_list(items):
    """A list of non-empty flat dicts (`details`) at depth 2: ONE call of the C encoder with the innermost
    indentation in the item separator; the element boundaries "},<newline + 16 spaces>{" (a literal line break cannot
    occur inside a JSON string) are then re-indented to the list's level."""
    if not items:
        return "[]"
    body = json.dumps(items, sort_keys=True, separators=(",\n" + _P16, ": "))
    body = body.replace("},\n" + _P16 + "{", "\n" + _P12 + "},\n" + _P12 + "{\n" + _P16)
    return "[\n" + _P12 + "{\n" + _P16 + body[2:-2] + "\n" + _P12 + "}\n" + _P8 + "]"


def dumps_result(results):
    """json.dumps(results, sort_keys=True, indent=4, separators=(',', ': ')) for a run() result, byte for byte, many
    times faster: with an indent the standard encoder runs in pure Python, and a sample's `details` alone are
    thousands of strings.  The structure is known -- {samplekey, bam, tredCalls: {key: scalar | flat dict | list of
    flat dicts}} -- so: all scalar entries go through the C encoder in ONE call (the item separator carries the line
    break and the indentation; a literal line break cannot occur inside a JSON string, so the text splits back into
    one line per sorted key), every flat dict in one call, `details` natively from the scan's pools."""
    calls = results["tredCalls"]
    sep = ",\n" + _P8
    # (exact types first: an isinstance() against the Mapping / Sequence views costs an ABC check per key, 675 keys)
    kinds = {k: type(v) for k, v in calls.items()}
    plain = (int, float, str, bool, type(None))
    scalars = {k: v for k, v in calls.items()
               if kinds[k] in plain or not isinstance(v, (dict, list, Details, SparseDist))}
    entry = {}
    if scalars:
        lines = json.dumps(scalars, sort_keys=True, separators=(sep, ": "))[1:-1].split(sep)
        entry = dict(zip(sorted(scalars), lines))
    # all distributions of the sample in one native call, all `details` lists (they share the sample's scan) in another
    native = {}
    dist_keys = [k for k, t in kinds.items() if t is SparseDist]
    if dist_keys:
        texts = bamio.sparse_json_many([(calls[k].a, calls[k].b, calls[k].values) for k in dist_keys], 2)
        if texts is not None:
            native.update(zip(dist_keys, texts))
    det_keys = [k for k, t in kinds.items() if t is Details]
    if det_keys and all(calls[k].scan is calls[det_keys[0]].scan for k in det_keys):
        sc = calls[det_keys[0]].scan
        texts = bamio.details_json_many(sc.seq4, sc.seq4_off, sc.read_len, sc.name_blob, sc.name_off,
                                        [(calls[k].reads, calls[k].tags, calls[k].hs) for k in det_keys])
        if texts is not None:
            native.update(zip(det_keys, texts))
    for key, v in calls.items():
        if key in scalars:
            continue
        text = native.get(key)
        if text is not None:
            pass
        elif isinstance(v, dict):
            text = _flat(v, 2)
        elif isinstance(v, SparseDist):
            text = v.json_text(2) if key not in native else None
            if text is None:
                text = _flat(v.as_dict(), 2)
        elif isinstance(v, Details):
            text = v.json_text() if key not in native else None
            if text is None:
                text = _flat_list(v.items())
        else:
            text = _flat_list(v)
        entry[key] = json.dumps(key) + ": " + text
    inner = "{\n" + _P8 + sep.join(entry[k] for k in sorted(entry)) + "\n    }" if entry else "{}"
    top = {k: v for k, v in results.items() if k != "tredCalls"}
    parts = [(k, json.dumps(v)) for k, v in top.items()] + [("tredCalls", inner)]
    return "{\n" + ",\n".join("    " + json.dumps(k) + ": " + t for k, t in sorted(parts)) + "\n}"


def to_json(results, ref=None, repo=None, treds=None, store=None, quiet=False):
    """<samplekey>.json in the working directory (and on stdout): sorted keys, 4-space indent."""
    if not results["tredCalls"]:
        return
    text = dumps_result(results)
    if not quiet:
        print(text)
    with open(results["samplekey"] + ".json", "wb") as fw:      # (bytes: one encode, no text layer per write)
        fw.write((text + "\n").encode("utf-8"))


def vcfstanza(sampleid, bam, calls, ref):
    today = date.today()
    head = ["##fileformat=VCFv4.1", "##fileDate={}{:02d}{:02d}".format(today.year, today.month, today.day),
            "##source={} {}".format(__file__, bam), "##reference={}".format(ref),
            "##inferredGender={} depthY={}".format(calls["inferredGender"], calls["depthY"]),
            "##readLen={}bp".format(calls["readLen"])]
    columns = "#" + "\t".join("CHROM POS ID REF ALT QUAL FILTER INFO FORMAT".split() + [sampleid])
    return "\n".join(head) + "\n" + INFO + columns


def _vcf_line(name, calls, site):
    """One locus as a VCF record.  GT is relative to the reference copy number: 0/0 both alleles reference, 0/1 one
    of them, 1/1 both the same non-reference size, 1/2 two different ones; GB carries the absolute sizes."""
    chrom, pos, ref_copy, motif, info = site
    a, b = calls[name + ".1"], calls[name + ".2"]
    novel = sorted({a, b} - {ref_copy})
    if not novel:
        gt = "0/0"
    else:
        info += ";RPA=" + ",".join(str(x) for x in novel)
        gt = "0/1" if ref_copy in (a, b) else ("1/1" if len(novel) == 1 else "1/2")
    alt = ",".join(motif * x for x in novel) if novel and novel[0] != -1 else "."
    sample = [gt, "{}/{}".format(a, b)] + [str(calls[name + "." + k]) for k in
                                           ("FR", "PR", "RR", "DP", "FDP", "PDP", "RDP", "PEDP", "CI")]
    sample += ["{:.4g}".format(calls[name + ".PP"]), str(calls[name + ".label"])]
    fields = (chrom, pos, name, motif * ref_copy, alt, ".", ".", info, _SAMPLE_KEYS, ":".join(sample))
    return chrom, pos, "\t".join(str(x) for x in fields)


def to_vcf(results, ref, repo, treds=("HD",), store=None):
    """<samplekey>.tred.vcf.gz: one record per called locus, ordered by contig name and position."""
    calls = results["tredCalls"]
    if not calls:
        return
    records = sorted(_vcf_line(t, calls, repo.get_info(t)) for t in treds if t + ".1" in calls)
    text = vcfstanza(results["samplekey"], results["bam"], calls, ref) + "\n" + "".join(line + "\n" for _, _, line in records)
    with open(results["samplekey"] + ".tred.vcf.gz", "wb") as fw:   # (one gzip member written in one piece)
        fw.write(gzip.compress(text.encode("utf-8"), compresslevel=3))   # (as the native writer: emit.cpp GZIP_LEVEL)


def write_vcf_json(results, ref, repo, treds, store=None, quiet=False):
    try:
        to_vcf(results, ref, repo, treds=treds, store=store)
        to_json(results, ref, repo, treds=treds, store=store, quiet=quiet)
    except Exception as e:
        print("Error writing: {} ({})".format(results.get("samplekey"), e), file=sys.stderr)


# ---- driver -------------------------------------------------------------------------------------------------------
def _fan_out(argv, n_gpus, samples, launch_dir, no_output, quiet, devices=None, drivers=1, gpu=0):
    """--gpus N x --drivers D: this process stays off the GPU and starts D driver processes per GPU
    (shard.spawn_ranks, rank r on GPU r mod N) from the directory the command was given in; child r genotypes the samples
    the parent assigns it (balanced by BAM size) -- the list is fixed here and handed over in a file, so that every child
    partitions the same list -- and writes those samples' files.  Afterwards the JSONs are echoed in sample order."""
    import tempfile
    from . import shard
    if devices is None:
        devices = shard.visible_gpus()
    if devices < 1:
        raise SystemExit("tred.py: no GPU visible")
    devices = min(devices, n_gpus)          # the drivers share --gpus devices, not every device of the box
    n_gpus = n_gpus * max(1, drivers)       # (ranks from here on)
    samplekeys = [s[0] for s in samples]
    env = dict(os.environ)
    if devices == 1 and gpu:                # --gpus 1 --gpu K with several drivers: all of them on device K
        env["HIP_VISIBLE_DEVICES"] = shard.device_entry(gpu, env)
        env.pop("CUDA_VISIBLE_DEVICES", None)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                        [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    # which rank takes which sample: by BAM size, so that a cohort of mixed coverage keeps all GPUs busy to the end
    def size_of(path):
        try:
            return os.path.getsize(path)
        except OSError:
            return 0
    owners = shard.balanced_owners([size_of(s[1]) for s in samples], n_gpus)
    with tempfile.NamedTemporaryFile("w", suffix=".json", prefix="tred_tasks_", delete=False) as fp:
        json.dump({"samples": [list(s) for s in samples], "owner": owners}, fp)
    try:
        # --cleanup is the parent's: a child removing the shared working directory would take its siblings' files
        child_argv = [a for a in argv if a != "--cleanup"]
        cmd = [sys.executable, "-m", "tredparse_amd.tred"] + child_argv + ["--task-file", fp.name]
        codes = shard.spawn_ranks(cmd, n_gpus, devices, env=env, cwd=launch_dir)
    finally:
        os.unlink(fp.name)
    if not (no_output or quiet):
        # (the files' bytes as they are: no decoding and re-encoding of 300 KB per sample)
        out = getattr(sys.stdout, "buffer", None)
        sys.stdout.flush()
        for key in samplekeys:
            if os.path.exists(key + ".json"):
                if out is not None:
                    with open(key + ".json", "rb") as fp:
                        out.write(fp.read())
                else:
                    with open(key + ".json") as fp:
                        sys.stdout.write(fp.read())
        if out is not None:
            out.flush()
    return max(codes) if codes else 0


def default_cpus(gpus, pinned, usable=None):
    """--cpus when it is not given: the reference starts cpu_count() workers (tredparse/tred.py:88); here the CPUs this
    process may really keep busy (affinity mask, cgroup quota) shared among the ranks, and in a rank never more
    than its CPU set holds."""
    from . import shard
    share = max(1, (shard.usable_cpus() if usable is None else usable) // max(1, gpus))
    return max(1, min(share, len(pinned))) if pinned else share


def plan_drivers(args, n_samples, usable):
    """(driver processes per GPU, scan threads per driver) of this run: --drivers / --cpus where given, else
    shard.driver_plan's rule -- with no more drivers than there are blocks of 32 samples for (a driver costs seconds
    of start-up: interpreter, HIP context, pinned staging)."""
    from . import shard
    gpus = max(1, args.gpus)
    if str(args.drivers) != "auto":
        per_gpu = max(1, int(args.drivers))
        threads = args.cpus or max(1, usable // (per_gpu * gpus))
        return per_gpu, threads
    per_gpu, threads = shard.driver_plan(usable, gpus, gpu_inflate=args.gpu_inflate)
    fit = max(1, n_samples // (32 * gpus))
    if per_gpu > fit:
        per_gpu = fit
        threads = max(1, min(usable // (per_gpu * gpus), 16))
    return per_gpu, args.cpus or threads


def main(args, quiet=False):
    argv = list(args)
    p = set_argparse()
    args = p.parse_args(argv)
    quiet = quiet or args.quiet_json
    from . import shard
    usable = shard.usable_cpus()               # (of the inherited mask: what the whole job has, before this rank is pinned)
    pinned = shard.apply_rank_cpuset() if args.task_file else None   # a --gpus child: its GPU's NUMA node, before any GPU call
    if args.cpus is None and args.task_file:
        args.cpus = default_cpus(int(os.environ.get("WORLD_SIZE", "1")), pinned, usable)
    elif args.cpus is not None and pinned:
        args.cpus = max(1, min(args.cpus, len(pinned)))         # never more scan threads than this rank's CPU set holds
    if args.batch_samples is None:
        # (with the selection on the device a chunk is one decode + walk + select call and one genotyping call, nothing comes
        #  back but results: 36 samples per call fill the device -- 55.8 / 60.6 / 64.0 k genotypes/s at 12 / 24 / 36 --, at
        #  1.3 GB of pinned staging per driver)
        args.batch_samples = (36 if (args.gpu_walk and args.gpu_select) else 12) if args.gpu_inflate else 64
    logger.setLevel(getattr(logging, args.log))
    logging.getLogger("tredparse_amd.bam").setLevel(getattr(logging, args.log))
    t0 = time.time()
    infile = args.input_bam_path or args.infile
    if not infile:
        p.print_help()
        sys.exit(1)
    owners = None
    if args.task_file:                         # a --gpus child: the parent's sample list (and who takes what)
        with open(args.task_file) as fp:
            plan = json.load(fp)
        if isinstance(plan, dict):
            samples, owners = [tuple(x) for x in plan["samples"]], plan.get("owner")
        else:
            samples = [tuple(x) for x in plan]
    else:
        samples = read_csv(infile, args)       # paths become absolute here, before the working directory changes
    cwd = os.getcwd()
    sites = os.path.join(cwd, "sites")
    os.makedirs(args.workdir, exist_ok=True)
    os.chdir(args.workdir)
    try:
        repo = TREDsRepo(ref=args.ref, toy=args.toy, sites=sites)
        repo.set_ploidy(args.haploid)
        loci = ["HD"] if args.toy else (args.tred or repo.names)
        spawned = bool(args.task_file)
        if args.checkexists and not spawned:
            samples = [s for s in samples if not os.path.exists(s[0] + ".json")]
        tasks = [(key, bam, repo, [only] if only else loci, args.maxinsert, args.fullsearch, args.useclippedreads,
                  not args.noalts, not args.norepeatpairs, args.log) for key, bam, only in samples]
        if not tasks:
            return
        drivers = 1
        if not spawned:
            drivers, args.cpus = plan_drivers(args, len(tasks), usable)
        if args.gpus * drivers > 1 and not spawned:
            rc = _fan_out(argv + ["--cpus", str(args.cpus)], args.gpus, samples, cwd, args.no_output, quiet, drivers=drivers,
                          gpu=args.gpu)
            if rc:
                sys.exit(rc)
        else:
            device = args.gpu
            if spawned:                            # one of the --gpus children: its share of the samples, device 0
                from .shard import shard_range
                rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
                if owners is not None and len(owners) == len(tasks):
                    tasks = [t for t, o in zip(tasks, owners) if o == rank]
                else:
                    lo, hi = shard_range(len(tasks), rank, world)
                    tasks = tasks[lo:hi]
                device, quiet = 0, True
            if tasks:
                from .engine import Engine
                mark("main")
                engine = Engine(device)
                mark("engine")

                # the samples' files are written natively, from the batch's arrays, on threads of their own (Emitter); a
                # sample the native printers do not cover takes the Python path there
                emit = Emitter(args.ref, repo, loci, no_output=args.no_output, echo=not quiet,
                               workers=(3 if (args.gpu_inflate and args.gpu_walk and args.gpu_select) else 2) if args.cpus > 1 else 1)
                try:
                    run_many(tasks, engine, batch=max(1, args.batch_samples), threads=max(1, args.cpus), lazy_details=True,
                             inflate_device=device if (args.gpu_inflate and (args.cpus > 1 or (args.gpu_walk and args.gpu_select))) else None,
                             gpu_walk=args.gpu_walk,
                             gpu_select=args.gpu_select, emit=emit)
                    mark("run_many returned")
                finally:
                    emit.close()
                    mark("emitter closed")
                    # the pinned staging and the context are given back HERE, in order: left to the interpreter's shutdown
                    # (atexit, finalisers in any order) the same work took 0.85 s of a 12 288-sample command's 9.2 instead of 0.2
                    # (a driver process of the fan-out, which ends here; a caller of main() in a longer-lived process keeps both)
                    if spawned:
                        from .feeder import release_inflaters
                        release_inflaters()
                        mark("inflaters released")
                        engine.close()
                        mark("engine closed")
                    timeline_dump()
        print("Elapsed time={}".format(timedelta(seconds=time.time() - t0)), file=sys.stderr)
    finally:
        os.chdir(cwd)
    if args.cleanup and not args.task_file:        # a --gpus child never removes the directory its siblings write to
        shutil.rmtree(args.workdir, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1:])
