"""What the pieces of a driver process share (tred.py the CLI and run_many, feeder.py the GPU-inflate pipeline, emit.py
the native writer): the stage timers and the named form of run()'s argument tuple."""
import json
import os
import threading
import time

from .bam_parser import scan_sample


# seconds accumulated over run_many calls: the driver thread's waits for scans, its GPU calls and its formatting; and
# the writer thread's time in the sink (JSON / VCF text and files)
TIMING = {"scan_wait": 0.0, "gpu": 0.0, "format": 0.0, "write": 0.0, "inflate": 0.0, "inflate_blocks": 0, "inflate_failed": 0,
          "inflate_hits": 0, "inflate_misses": 0, "inflate_gpu": 0.0, "walk_regions": 0, "walk_declined": 0,
          "walk_blocks_fetched": 0, "walk_alt_regions": 0, "walk_alt_declined": 0, "walk_call": 0.0, "walk_fetch": 0.0, "pack": 0.0,
          "merged_chunks": 0, "select_samples": 0, "select_declined": 0, "gpu_calls": 0}
_TIMING_LOCK = threading.Lock()


def timing_add(**kw):
    """TIMING[key] += value for every keyword, under a lock: the scan pool, the feeder and the writer thread all
    report here while the driver thread does too (a lost update would show up in bench.py's driver_seconds)."""
    with _TIMING_LOCK:
        for k, v in kw.items():
            TIMING[k] += v


# A driver's timeline, for tuning: with TRED_TIMELINE=<directory> in the environment every mark(event) is kept with its wall-clock
# time and the process writes <directory>/timeline_<pid>.json when it ends (tools/cli_rate.py --timeline reads them).  Off: one
# dictionary look-up per mark.
_TIMELINE = [] if os.environ.get("TRED_TIMELINE") else None


def mark(event, **kw):
    if _TIMELINE is not None:
        _TIMELINE.append((time.time(), event, kw))


def timeline_dump():
    if _TIMELINE:
        try:
            with open(os.path.join(os.environ["TRED_TIMELINE"], "timeline_{}.json".format(os.getpid())), "w") as fp:
                json.dump(_TIMELINE, fp)
        except OSError:
            pass


def _options(arg):
    """The reference's run() argument tuple, named."""
    samplekey, bam, repo, names, maxinsert, fullsearch, clip, alts, repeatpairs, log = arg
    return dict(samplekey=samplekey, bam=bam, repo=repo, names=list(names), maxinsert=maxinsert,
                fullsearch=fullsearch, clip=clip, alts=alts, repeatpairs=repeatpairs, log=log)


def collect_sample(arg):
    """Host half of a sample (thread-safe, no GPU): the native scan of its BAM."""
    o = _options(arg)
    return scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"])
