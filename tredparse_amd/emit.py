"""The native writer's driver side: a sample's <key>.json and <key>.tred.vcf.gz straight from a genotyped batch's arrays
(libtredbam.so tredbam_emit_sample_files, csrc/emit.cpp).  Reference counterpart: the tail of run() and to_json / to_vcf,
tredparse/tred.py:251-275, 296-374."""
import logging
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import date

from .runtime import _options, timing_add

logger = logging.getLogger("tredparse_amd.tred")

NATIVE_EMIT = True        # (tools/prof_host.py and the tests switch the native writer off to time / compare the Python path)


class Emitter(object):
    """Writes the samples' <key>.json and <key>.tred.vcf.gz straight from a batch's result arrays and the scans' pools,
    natively (libtredbam.so tredbam_emit_sample_files, include/tredbam.h) on `workers` threads that run WITHOUT the
    interpreter lock -- instead of building every sample's tredCalls dict (format_scans) and printing it (to_json,
    to_vcf) in Python, which was what bounded a driver process.  The text is byte for byte the Python path's
    (tests/test_emit_native.py).  A sample the native printers do not cover -- --log DEBUG, a BAM that did not open, a
    batch the retries cut into single units, names outside ASCII -- goes through the Python path on the same thread.
      echo      print each JSON on stdout as to_json does (one worker then: the order of the samples is kept)
      on_sample called with {'samplekey', 'names', 'printed' (bool per locus), 'first_allele' (units, per locus)} after a
                sample's files are written (bench.py checks the calls against the simulated alleles with it)"""

    def __init__(self, ref, repo, treds, no_output=False, echo=False, workers=2, on_sample=None, depth=96):
        from . import bamio
        self.ref, self.repo, self.treds, self.no_output, self.echo, self.on_sample = ref, repo, list(treds), no_output, echo, on_sample
        self.lib = bamio._native() if NATIVE_EMIT else None
        self.tables = {}
        self.error = None
        self.pool = ThreadPoolExecutor(max_workers=1 if echo else max(1, workers), thread_name_prefix="tred-emit")
        self.depth = depth
        self.room = threading.BoundedSemaphore(depth)          # results in flight (each holds its scan and batch arrays)
        from . import tred as _tred               # (the VCF header's ##source names the command-line module, as the Python writer does)
        self.meta = _tred.INFO.encode("utf-8")
        self.source = _tred.__file__.encode("utf-8")

    def _table(self, names):
        from . import bamio
        key = tuple(names)
        if key not in self.tables:
            ok = self.lib is not None and all(isinstance(self.repo[n].cutoff_risk, int) and isinstance(self.repo[n].cutoff_prerisk, int)
                                              for n in names)
            self.tables[key] = bamio.emit_locus_table(self.repo, names) if ok else None
        return self.tables[key]

    def submit(self, arg, scan, pieces):
        """One sample of a genotyped batch (pieces: genotype_scans' parts for it)."""
        self.room.acquire()
        try:
            self.pool.submit(self._run, arg, scan, pieces)
        except BaseException:
            self.room.release()
            raise

    def _run(self, arg, scan, pieces):
        t0 = time.perf_counter()
        try:
            if self.error is None:
                self._emit(arg, scan, pieces)
        except BaseException as e:
            self.error = e
        finally:
            self.room.release()
            timing_add(write=time.perf_counter() - t0)

    def _python_path(self, arg, scan, pieces):
        """The sample through format_scans and the Python writers (what the native path must equal)."""
        from .tred import format_scans, unit_results, write_vcf_json
        picks = [(0, scan, [k for _, _, ks in pieces for k in ks])]
        result = format_scans([arg], [scan], picks, unit_results({0: pieces}), lazy_details=True)[0]
        if not self.no_output:
            write_vcf_json(result, self.ref, self.repo, self.treds, quiet=not self.echo)
        if self.on_sample is not None:
            calls = result["tredCalls"]
            self.on_sample({"samplekey": result["samplekey"], "names": scan.names,
                            "printed": [n + ".1" in calls for n in scan.names],
                            "first_allele": [calls.get(n + ".1", -1) for n in scan.names]})

    def _emit(self, arg, scan, pieces):
        import ctypes as C
        import numpy as np
        from . import bamio
        o = _options(arg)
        native = scan.opened and len(pieces) == 1 and o["log"] != "DEBUG" and getattr(pieces[0][0], "joint_units", None) is not None
        table = self._table(scan.names) if native else None
        if table is None or (self.no_output and self.on_sample is None):
            if not (self.no_output and self.on_sample is None and o["log"] != "DEBUG"):
                self._python_path(arg, scan, pieces)
            else:                # nothing to print: the loci the grid refused are still reported, as the Python path does
                from .models import STATUS_ERRORS
                for br, i0, ks in pieces:
                    st = br.calls["status"][i0:i0 + len(ks)]
                    for j in np.nonzero(st < 0)[0].tolist():
                        logger.error("Exception on `%s` %s (%s)", o["bam"], scan.names[ks[j]],
                                     STATUS_ERRORS.get(int(st[j]), "status {}".format(int(st[j]))))
            return
        br, i0, ks = pieces[0]
        eb = getattr(br, "_emit", None)
        if eb is None:                       # the batch's arrays, once per batch (whichever sample's thread gets there first)
            a, b, v, lo, n = br.joint_units
            keep = [np.ascontiguousarray(x) for x in (br.tag, br.h, br.batch.unit_read_off, br.calls, br.marg, a, b, v, lo, n)]
            eb = bamio.EmitBatch(*[x.ctypes.data for x in keep[:5]], br.marg.shape[2], *[x.ctypes.data for x in keep[5:]],
                                 int(bool(br.repeatpairs)), 0)
            eb.keep = keep
            br._emit = eb
        index = np.full(len(scan.names), -1, np.int32)
        index[ks] = np.arange(i0, i0 + len(ks), dtype=np.int32)
        depth = np.ascontiguousarray(scan.depth, np.float64)
        key, bam = o["samplekey"].encode("utf-8"), o["bam"].encode("utf-8")
        ydepth = float(scan.ydepth) if isinstance(scan.ydepth, float) else -1.0
        es = bamio.EmitSample(key, bam, scan.gender.encode("utf-8"), ydepth, 1, int(scan.readlen),
                              scan.seq4.ctypes.data, scan.seq4_off.ctypes.data, scan.read_len.ctypes.data,
                              scan.name_blob if isinstance(scan.name_blob, int) else C.cast(C.c_char_p(scan.name_blob), C.c_void_p).value,
                              scan.name_off.ctypes.data, scan.name_id.ctypes.data,
                              scan.global_lens.ctypes.data, scan.target_lens.ctypes.data, scan.unit.ctypes.data, depth.ctypes.data,
                              index.ctypes.data)
        today = date.today()
        eo = bamio.EmitOpts(self.ref.encode("utf-8"), self.source, "{}{:02d}{:02d}".format(today.year, today.month, today.day).encode(),
                            self.meta, 0 if self.no_output else 1, 0 if self.no_output else 1, 0, 0)     # (gzip level 0: the writer's own, emit.cpp GZIP_LEVEL)
        status = np.zeros(max(1, len(scan.names)), np.int32)
        cap = (1 << 22) if self.echo else 0
        text = C.create_string_buffer(cap) if cap else None
        got = C.c_int64(-1)
        rc = self.lib.tredbam_emit_sample_files(C.addressof(table), len(scan.names), C.addressof(eb), C.addressof(es), C.addressof(eo),
                                                status.ctypes.data, text, cap, C.byref(got))
        if rc == 1:
            return self._python_path(arg, scan, pieces)
        if rc < 0:
            why = self.lib.tredbam_emit_last_error().decode("utf-8", "replace") or str(rc)
            print("Error writing: {} ({})".format(o["samplekey"], why), file=sys.stderr)
            if rc != -5:         # not an I/O error of this one sample's files (the reference prints and goes on there, tred.py:
                # 290-293): the arrays handed over do not fit together -- every later sample of the run would be wrong as well
                raise RuntimeError("native writer refused `{}`: {} (rc={})".format(o["samplekey"], why, rc))
            if self.on_sample is not None:
                self.on_sample({"samplekey": o["samplekey"], "names": scan.names, "printed": [False] * len(scan.names),
                                "first_allele": [-1] * len(scan.names)})
            return
        from .models import STATUS_ERRORS
        for k in np.nonzero(status[:len(scan.names)] < 0)[0].tolist():
            st = int(status[k])
            logger.error("Exception on `%s` %s (%s)", o["bam"], scan.names[k], STATUS_ERRORS.get(st, "status {}".format(st)))
        if self.echo and not self.no_output:
            if got.value >= 0:
                print(text.raw[:got.value].decode("ascii"))
            else:
                with open(o["samplekey"] + ".json") as fp:
                    sys.stdout.write(fp.read())
        if self.on_sample is not None:
            c = br.calls[i0:i0 + len(ks)]
            per = np.array([len(scan.loci[k].repeat) for k in ks], np.int64)
            first = np.full(len(scan.names), -1, np.int64)
            first[ks] = np.where(c["status"] == 0, np.minimum(c["h1"], c["h2"]) // per, -1)
            self.on_sample({"samplekey": o["samplekey"], "names": scan.names, "printed": (status[:len(scan.names)] == 0).tolist(),
                            "first_allele": first.tolist()})

    def drain(self):
        """Waits until every submitted sample is written (a sample holds one of the `depth` places until it is)."""
        for _ in range(self.depth):
            self.room.acquire()
        for _ in range(self.depth):
            self.room.release()
        if self.error is not None:
            raise self.error

    def close(self):
        self.pool.shutdown(wait=True)
        if self.error is not None:
            raise self.error
