"""ctypes binding of libtredgpu.so (include/tredgpu.h).

The library is the product path: there is no Python/CPU fallback.  Loading needs the in-tree
``tredparse_amd/libtredgpu.so`` (built by ``__graft_entry__.build()`` / ``make -C tredparse_amd/csrc``);
creating a context needs a HIP device and fails loudly otherwise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TREDGPU_LIB") or os.path.join(HERE, "libtredgpu.so")   # override: kernel experiments

MEM_HOST, MEM_DEVICE = 0, 1
TAG_NONE, TAG_FULL, TAG_PREF, TAG_POST, TAG_REPT, TAG_HANG, TAG_INVALID = 0, 1, 2, 3, 4, 5, 255
TAG_NAMES = {TAG_FULL: "FULL", TAG_PREF: "PREF", TAG_POST: "POST", TAG_REPT: "REPT", TAG_HANG: "HANG"}
SPAN = 1000


class TredGpuError(RuntimeError):
    pass


class SwParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("flank", C.c_int32), ("clip", C.c_int32),
                ("max_read_len", C.c_int32), ("reserved", C.c_int32)]


class UnitParams(C.Structure):
    _fields_ = [("period", C.c_int32), ("readlen", C.c_int32), ("ploidy", C.c_int32),
                ("maxinsert", C.c_int32), ("fullsearch", C.c_int32), ("ref_len", C.c_int32),
                ("minpe", C.c_int32), ("cutoff_risk", C.c_int32), ("is_expansion", C.c_int32),
                ("is_recessive", C.c_int32), ("pe_off", C.c_int32), ("n_global", C.c_int32),
                ("tl_off", C.c_int32), ("n_target", C.c_int32), ("half_depth", C.c_double)]


class Call(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_pairs", C.c_int32), ("h1", C.c_int32), ("h2", C.c_int32),
                ("ci", C.c_int32 * 4), ("run_pe", C.c_int32), ("pad", C.c_int32),
                ("lik", C.c_double), ("pp", C.c_double)]


UNIT_DTYPE = np.dtype([("period", "<i4"), ("readlen", "<i4"), ("ploidy", "<i4"), ("maxinsert", "<i4"),
                       ("fullsearch", "<i4"), ("ref_len", "<i4"), ("minpe", "<i4"), ("cutoff_risk", "<i4"),
                       ("is_expansion", "<i4"), ("is_recessive", "<i4"), ("pe_off", "<i4"),
                       ("n_global", "<i4"), ("tl_off", "<i4"), ("n_target", "<i4"), ("half_depth", "<f8")])
CALL_DTYPE = np.dtype([("status", "<i4"), ("n_pairs", "<i4"), ("h1", "<i4"), ("h2", "<i4"),
                       ("ci", "<i4", (4,)), ("run_pe", "<i4"), ("pad", "<i4"), ("lik", "<f8"), ("pp", "<f8")])
assert UNIT_DTYPE.itemsize == C.sizeof(UnitParams) == 64
assert CALL_DTYPE.itemsize == C.sizeof(Call) == 56

# every symbol include/tredgpu.h declares (tests check the .so exports them all)
EXPORTS = ("tredgpu_create", "tredgpu_destroy", "tredgpu_last_error", "tredgpu_sync", "tredgpu_get_stream",
           "tredgpu_version", "tredgpu_set_ladders", "tredgpu_set_model", "tredgpu_pack_reads",
           "tredgpu_sw_classify", "tredgpu_tally", "tredgpu_likelihood_grid", "tredgpu_likelihood_grid_joint",
           "tredgpu_genotype_batch", "tredgpu_genotype_batch_joint", "tredgpu_genotype_selected",
           "tredgpu_pe_kde", "tredgpu_reset_timing", "tredgpu_get_timing", "tredgpu_get_sw_counters",
           "tredgpu_inflater_create", "tredgpu_inflater_destroy", "tredgpu_inflater_last_error",
           "tredgpu_inflater_reserve", "tredgpu_inflate_blocks", "tredgpu_inflate_blocks_crc", "tredgpu_inflater_timing",
           "tredgpu_inflate_walk", "tredgpu_inflater_fetch", "tredgpu_inflater_walk_ms", "tredgpu_inflater_host_out",
           "tredgpu_inflater_fetch_dense", "tredgpu_inflater_pinned_bytes", "tredgpu_inflater_walk_serial_regions")

_lib = None


def load():
    """dlopen the in-tree library and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TredGpuError("{} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.tredgpu_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.tredgpu_destroy.argtypes = [vp]
    lib.tredgpu_destroy.restype = None
    lib.tredgpu_last_error.argtypes = [vp]
    lib.tredgpu_last_error.restype = C.c_char_p
    lib.tredgpu_sync.argtypes = [vp]
    lib.tredgpu_get_stream.argtypes = [vp]
    lib.tredgpu_get_stream.restype = vp
    lib.tredgpu_version.restype = C.c_char_p
    lib.tredgpu_set_ladders.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_char_p), vp]
    lib.tredgpu_set_model.argtypes = [vp, vp, vp, C.c_double, C.c_double]
    lib.tredgpu_pack_reads.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.tredgpu_pack_reads.restype = i64
    lib.tredgpu_sw_classify.argtypes = [vp, C.c_int, vp, vp, vp, i64, vp, vp, i32, C.POINTER(SwParams),
                                        vp, vp, vp, vp, i32]
    lib.tredgpu_tally.argtypes = [vp, C.c_int, vp, vp, i64, vp, i32, vp, i32, vp, vp, vp]
    lib.tredgpu_likelihood_grid.argtypes = [vp, C.c_int, vp, i32, i32, vp, vp, vp, vp, i64, vp, i64, vp,
                                            vp, vp, vp, i32]
    lib.tredgpu_likelihood_grid_joint.argtypes = [vp, C.c_int, vp, i32, i32, vp, vp, vp, vp, i64, vp, i64, vp,
                                                  vp, i32, vp, vp, vp, vp]
    lib.tredgpu_genotype_batch.argtypes = [vp, C.c_int, vp, vp, vp, i64, vp, vp, vp, i32,
                                           C.POINTER(SwParams), vp, vp, i64, vp, i64, vp, vp, vp, i32,
                                           vp, vp, vp, vp]
    lib.tredgpu_genotype_batch_joint.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, C.POINTER(SwParams), vp, vp, i64, vp, i64,
                                                 vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.tredgpu_genotype_selected.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(SwParams), vp, i64, vp, i64,
                                              vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tredgpu_pe_kde.argtypes = [vp, C.c_int, vp, i32, vp, i64, vp, vp]
    lib.tredgpu_reset_timing.argtypes = [vp]
    lib.tredgpu_get_timing.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(C.c_double)]
    lib.tredgpu_get_sw_counters.argtypes = [vp, vp]
    lib.tredgpu_inflater_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.tredgpu_inflater_destroy.argtypes = [vp]
    lib.tredgpu_inflater_destroy.restype = None
    lib.tredgpu_inflater_last_error.argtypes = [vp]
    lib.tredgpu_inflater_last_error.restype = C.c_char_p
    lib.tredgpu_inflater_reserve.argtypes = [vp, i64, i64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.tredgpu_inflate_blocks.argtypes = [vp, i32, vp]
    lib.tredgpu_inflate_blocks_crc.argtypes = [vp, i32, vp, vp]
    lib.tredgpu_inflater_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.tredgpu_inflate_walk.argtypes = [vp, i32, vp, vp, C.POINTER(WalkArgs)]
    lib.tredgpu_inflater_fetch.argtypes = [vp, i32, vp]
    lib.tredgpu_inflater_walk_ms.argtypes = [vp, C.POINTER(C.c_double)]
    lib.tredgpu_inflater_host_out.argtypes = [vp, C.c_int]
    lib.tredgpu_inflater_pinned_bytes.argtypes = [vp]
    lib.tredgpu_inflater_pinned_bytes.restype = i64
    lib.tredgpu_inflater_walk_serial_regions.argtypes = [vp]
    lib.tredgpu_inflater_walk_serial_regions.restype = i64
    lib.tredgpu_inflater_fetch_dense.argtypes = [vp, i32, vp, C.POINTER(vp), vp]
    _lib = lib
    return lib


def _ptr(a):
    """Raw pointer of a numpy array, a torch tensor, an int address or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return a.data_ptr()
    raise TypeError("unsupported buffer type {}".format(type(a)))


def pack_reads(seqs):
    """2-bit + N-mask packing (tredgpu_pack_reads).  seqs: list of str/bytes.
    Returns (packed uint32[], word_off int64[n+1], read_len int32[n])."""
    lib = load()
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    n = len(bs)
    off = np.zeros(n + 1, np.int64)
    if n:
        off[1:] = np.cumsum([len(b) for b in bs])
    blob = b"".join(bs)
    buf = np.frombuffer(blob, np.uint8) if blob else np.zeros(1, np.uint8)
    woff = np.zeros(n + 1, np.int64)
    rlen = np.zeros(max(n, 1), np.int32)
    total = lib.tredgpu_pack_reads(buf.ctypes.data, off.ctypes.data, n, None, woff.ctypes.data, rlen.ctypes.data)
    if total < 0:
        raise TredGpuError("tredgpu_pack_reads failed ({})".format(total))
    packed = np.zeros(max(int(total), 1), np.uint32)
    lib.tredgpu_pack_reads(buf.ctypes.data, off.ctypes.data, n, packed.ctypes.data, woff.ctypes.data, rlen.ctypes.data)
    return packed, woff, rlen[:n]


def pack_codes(codes, lengths=None):
    """Vectorised packer for a 2-D uint8 array of base codes (0..3, 4 = N), all reads the same
    length (rows) -- same record layout as tredgpu_pack_reads; used by the synthetic generator."""
    codes = np.ascontiguousarray(codes, np.uint8)
    n, L = codes.shape
    nb, nm = (L + 15) // 16, (L + 31) // 32
    isn = codes >= 4
    c = np.where(isn, 0, codes).astype(np.uint8)
    bits = np.zeros((n, nb * 32), np.uint8)          # bit 2k = low bit of base k, bit 2k+1 = high bit
    bits[:, 0:2 * L:2] = c & 1
    bits[:, 1:2 * L:2] = c >> 1
    words = np.packbits(bits, axis=1, bitorder="little").view("<u4")
    mbits = np.zeros((n, nm * 32), np.uint8)
    mbits[:, :L] = isn
    masks = np.packbits(mbits, axis=1, bitorder="little").view("<u4")
    packed = np.ascontiguousarray(np.concatenate([words, masks], axis=1).reshape(-1))
    woff = (np.arange(n + 1, dtype=np.int64) * (nb + nm))
    rlen = np.full(n, L, np.int32)
    return packed, woff, rlen


def version():
    """The library's version string incl. the hash of the kernel sources it was built from (csrc/Makefile)."""
    return load().tredgpu_version().decode()


class Context:
    """One GPU + one HIP stream (tredgpu_ctx)."""

    def __init__(self, device_id=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.tredgpu_create(device_id, C.byref(h))
        if rc != 0:
            raise TredGpuError("tredgpu_create({}) failed: {}".format(
                device_id, self.lib.tredgpu_last_error(None).decode()))
        self.h = h
        self.n_ladders = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tredgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise TredGpuError("{} failed ({}): {}".format(what, rc, self.lib.tredgpu_last_error(self.h).decode()))

    def sync(self):
        self._chk(self.lib.tredgpu_sync(self.h), "tredgpu_sync")

    @property
    def stream(self):
        return self.lib.tredgpu_get_stream(self.h)

    def set_ladders(self, ladders):
        """ladders: list of (prefix, repeat, suffix, max_units)."""
        n = len(ladders)
        arr = lambda k: (C.c_char_p * max(n, 1))(*[l[k].encode() for l in ladders])
        mu = np.asarray([l[3] for l in ladders] or [0], np.int32)
        self._chk(self.lib.tredgpu_set_ladders(self.h, n, arr(0), arr(1), arr(2), mu.ctypes.data),
                  "tredgpu_set_ladders")
        self.n_ladders = n
        self.ladders = list(ladders)

    def set_model(self, step_pdf, stutter_w, gc=.68, score=1.0):
        step = np.ascontiguousarray(step_pdf, np.float64)
        w = np.ascontiguousarray(stutter_w, np.float64)
        assert step.shape == (6, 37) and w.shape == (5,)
        self._chk(self.lib.tredgpu_set_model(self.h, step.ctypes.data, w.ctypes.data, gc, score),
                  "tredgpu_set_model")

    def sw_classify(self, mem, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, n_units,
                    params, out_tag, out_h, out_score, out_dump=None, dump_templates=0):
        self._chk(self.lib.tredgpu_sw_classify(self.h, mem, _ptr(packed), _ptr(read_off), _ptr(read_len),
                                               n_reads, _ptr(unit_read_off), _ptr(unit_ladder), n_units,
                                               C.byref(params), _ptr(out_tag), _ptr(out_h), _ptr(out_score),
                                               _ptr(out_dump), dump_templates), "tredgpu_sw_classify")

    def tally(self, mem, tag, h, n_reads, unit_read_off, n_units, read_pair_id, hist_stride, full_cnt,
              pref_cnt, rept_cnt):
        self._chk(self.lib.tredgpu_tally(self.h, mem, _ptr(tag), _ptr(h), n_reads, _ptr(unit_read_off),
                                         n_units, _ptr(read_pair_id), hist_stride, _ptr(full_cnt),
                                         _ptr(pref_cnt), _ptr(rept_cnt)), "tredgpu_tally")

    def likelihood_grid(self, mem, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens,
                        n_global_total, target_lens, n_target_total, calls, grid_off=None, grid_dump=None,
                        marg=None, marg_stride=0):
        self._chk(self.lib.tredgpu_likelihood_grid(self.h, mem, _ptr(units), n_units, hist_stride,
                                                   _ptr(full_cnt), _ptr(pref_cnt), _ptr(rept_cnt),
                                                   _ptr(global_lens), n_global_total, _ptr(target_lens),
                                                   n_target_total, _ptr(calls), _ptr(grid_off),
                                                   _ptr(grid_dump), _ptr(marg), marg_stride),
                  "tredgpu_likelihood_grid")

    def likelihood_grid_joint(self, mem, units, n_units, hist_stride, full_cnt, pref_cnt, rept_cnt, global_lens,
                              n_global_total, target_lens, n_target_total, calls, marg, marg_stride, joint_off, joint,
                              joint_n, joint_total):
        self._chk(self.lib.tredgpu_likelihood_grid_joint(self.h, mem, _ptr(units), n_units, hist_stride,
                                                         _ptr(full_cnt), _ptr(pref_cnt), _ptr(rept_cnt),
                                                         _ptr(global_lens), n_global_total, _ptr(target_lens),
                                                         n_target_total, _ptr(calls), _ptr(marg), marg_stride,
                                                         _ptr(joint_off), _ptr(joint), _ptr(joint_n), _ptr(joint_total)),
                  "tredgpu_likelihood_grid_joint")

    def genotype_batch(self, mem, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, units,
                       n_units, params, read_pair_id, global_lens, n_global_total, target_lens,
                       n_target_total, out_tag, out_h, out_score, hist_stride, full_cnt, pref_cnt, rept_cnt,
                       calls):
        self._chk(self.lib.tredgpu_genotype_batch(self.h, mem, _ptr(packed), _ptr(read_off), _ptr(read_len),
                                                  n_reads, _ptr(unit_read_off), _ptr(unit_ladder), _ptr(units),
                                                  n_units, C.byref(params), _ptr(read_pair_id),
                                                  _ptr(global_lens), n_global_total, _ptr(target_lens),
                                                  n_target_total, _ptr(out_tag), _ptr(out_h), _ptr(out_score),
                                                  hist_stride, _ptr(full_cnt), _ptr(pref_cnt), _ptr(rept_cnt),
                                                  _ptr(calls)), "tredgpu_genotype_batch")

    def genotype_batch_joint(self, packed, read_off, read_len, n_reads, unit_read_off, unit_ladder, units, n_units, params,
                             read_pair_id, global_lens, n_global_total, target_lens, n_target_total, out_tag, out_h, out_score,
                             hist_stride, rept_cnt, calls, marg, marg_stride, joint_off, joint, joint_n, joint_total):
        """tredgpu_genotype_batch_joint: SW + tagging -> histograms -> grid with marginals and sparse joint, host arrays,
        one wait."""
        self._chk(self.lib.tredgpu_genotype_batch_joint(self.h, _ptr(packed), _ptr(read_off), _ptr(read_len), n_reads,
                                                        _ptr(unit_read_off), _ptr(unit_ladder), _ptr(units), n_units,
                                                        C.byref(params), _ptr(read_pair_id), _ptr(global_lens), n_global_total,
                                                        _ptr(target_lens), n_target_total, _ptr(out_tag), _ptr(out_h),
                                                        _ptr(out_score), hist_stride, _ptr(rept_cnt), _ptr(calls), _ptr(marg),
                                                        marg_stride, _ptr(joint_off), _ptr(joint), _ptr(joint_n),
                                                        _ptr(joint_total)), "tredgpu_genotype_batch_joint")

    def genotype_selected(self, segs, unit_read_off, unit_word_off, unit_seq4_off, unit_name_off, unit_ladder, units, n_units, params,
                          global_lens, n_global_total, target_lens, n_target_total, out_tag, out_h, out_score, hist_stride, rept_cnt,
                          calls, marg, marg_stride, joint_off, joint, joint_n, joint_total, read_len, seq4_off, seq4, name_off, names):
        """tredgpu_genotype_selected: genotype_batch_joint over reads the inflaters' selections left on the device.
        segs: [(Inflater, int32 array of its tasks in batch order)]."""
        arr = (SelectedUnits * max(len(segs), 1))()
        keep = []
        for k, (inf, task) in enumerate(segs):
            task = np.ascontiguousarray(task, np.int32)
            keep.append(task)
            arr[k] = SelectedUnits(inf._h, len(task), 0, task.ctypes.data if len(task) else None)
        self._chk(self.lib.tredgpu_genotype_selected(self.h, arr, len(segs), _ptr(unit_read_off), _ptr(unit_word_off), _ptr(unit_seq4_off),
                                                     _ptr(unit_name_off), _ptr(unit_ladder), _ptr(units), n_units, C.byref(params),
                                                     _ptr(global_lens), n_global_total, _ptr(target_lens), n_target_total, _ptr(out_tag),
                                                     _ptr(out_h), _ptr(out_score), hist_stride, _ptr(rept_cnt), _ptr(calls), _ptr(marg),
                                                     marg_stride, _ptr(joint_off), _ptr(joint), _ptr(joint_n), _ptr(joint_total),
                                                     _ptr(read_len), _ptr(seq4_off), _ptr(seq4), _ptr(name_off), _ptr(names)),
                  "tredgpu_genotype_selected")

    def reset_timing(self):
        self._chk(self.lib.tredgpu_reset_timing(self.h), "tredgpu_reset_timing")

    def get_timing(self, which):
        """(launches, total device ms) of kernel `which` since reset_timing (HIP events)."""
        n, ms = C.c_int64(0), C.c_double(0)
        self._chk(self.lib.tredgpu_get_timing(self.h, which, C.byref(n), C.byref(ms)), "tredgpu_get_timing")
        return n.value, ms.value

    def get_sw_counters(self):
        """dict of the SW kernel's work counters since reset_timing (tredgpu_get_sw_counters)."""
        out = np.zeros(8, np.uint64)
        self._chk(self.lib.tredgpu_get_sw_counters(self.h, out.ctypes.data), "tredgpu_get_sw_counters")
        keys = ("trunk_cols", "continuation_cols", "templates_combined", "templates_dropped", "emitted_from_trunk", "waves", "read_cols")
        return {k: int(v) for k, v in zip(keys, out)}

    def pe_kde(self, mem, units, n_units, global_lens, n_global_total, pdf_out, status_out):
        self._chk(self.lib.tredgpu_pe_kde(self.h, mem, _ptr(units), n_units, _ptr(global_lens),
                                          n_global_total, _ptr(pdf_out), _ptr(status_out)), "tredgpu_pe_kde")


KERNEL_SW, KERNEL_TALLY, KERNEL_GRID = 0, 1, 2
KERNEL_GRID_PREPARE, KERNEL_GRID_PAIRS, KERNEL_GRID_REDUCE, KERNEL_GRID_KDE = 3, 4, 5, 6


def default_sw_params(clip=False, max_read_len=0):
    """bam_parser.py:95-98 scoring (1/5/7/2), FLANKMATCH 9 (bam_parser.py:30)."""
    return SwParams(1, 5, 7, 2, 9, int(bool(clip)), int(max_read_len), 0)


class WalkArgs(C.Structure):
    """tredgpu_walk_args (include/tredgpu.h)."""
    _fields_ = [("blk_coffset", C.c_void_p), ("blk_clen", C.c_void_p), ("blk_crc", C.c_void_p),
                ("tasks", C.c_void_p), ("n_tasks", C.c_int32), ("chunks", C.c_void_p), ("n_chunks", C.c_int32),
                ("results", C.c_void_p), ("global_pool", C.c_void_p), ("cap_global", C.c_int64),
                ("target_pool", C.c_void_p), ("cap_target", C.c_int64), ("n_global", C.c_int64), ("n_target", C.c_int64),
                ("alt_tasks", C.c_void_p), ("n_alt_tasks", C.c_int32), ("alt_chunks", C.c_void_p), ("n_alt_chunks", C.c_int32),
                ("alt_results", C.c_void_p), ("need", C.c_void_p), ("select", C.c_void_p), ("selected", C.c_void_p)]


class SelectedUnits(C.Structure):
    """tredgpu_selected_units (include/tredgpu.h section 5)."""
    _fields_ = [("inf", C.c_void_p), ("n_units", C.c_int32), ("pad", C.c_int32), ("task", C.c_void_p)]


# layouts of tredgpu_walk_task / _chunk / _result (the same as bamio.WALK_*_DTYPE: tredbam.h's structs)
WALK_TASK_DTYPE = np.dtype([(k, "<i4") for k in ("tid", "start", "end", "tstart", "tend", "span", "chunk_first", "n_chunks",
                                                 "block_first", "block_end", "win_lo", "win_hi")])
WALK_CHUNK_DTYPE = np.dtype([("begin_block", "<i4"), ("begin_upos", "<i4"), ("end_voffset", "<u8")])
WALK_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_global", "<i4"), ("n_target", "<i4"), ("n_window", "<i4"),
                              ("global_first", "<i8"), ("target_first", "<i8"), ("win_vbeg", "<u8"), ("win_vend", "<u8")])
ALT_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n", "<i4"), ("vbeg", "<u8", (6,))])
# tredgpu_select_task / tredgpu_select_result (section 5: the read selection on the device)
SELECT_TASK_DTYPE = np.dtype([("pos_lo", "<i4"), ("pos_hi", "<i4"), ("alt_first", "<i4"), ("n_alt", "<i4")])
SELECT_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_reads", "<i4"), ("n_words", "<i4"), ("seq4_bytes", "<i4"), ("name_bytes", "<i4"),
                                ("max_len", "<i4"), ("depth_sum", "<i8")])
assert SELECT_TASK_DTYPE.itemsize == 16 and SELECT_RESULT_DTYPE.itemsize == 32
SELECT_CAP = 4096


MIN_PAIR_BYTES = 2 * (36 + 2 + 4 + 18)    # two BAM records of a pair at their smallest: fixed fields, name, one CIGAR op, 36 bases


def walk_pool_pairs(tasks, out_off):
    """Upper bound of the pair lengths one walk call can produce -- the room its pools are given, shared by all tasks:
    every pair needs two records among the call's inflated bytes (out_off: the call's block offsets; a record of a read
    of >= 36 bases cannot be smaller than MIN_PAIR_BYTES / 2), and a byte belongs to the regions of at most two loci
    (FXS / FXTAS and SBMA / AR share their coordinates; loci further apart than +-10 kb share nothing).  30x samples
    sit near a tenth of this bound; a locus list with three or more loci on one spot could exceed it -- those regions
    then come back with status 6 and the host walks them."""
    if len(tasks) == 0:
        return 0
    return 2 * int(np.asarray(out_off, np.int64)[-1]) // MIN_PAIR_BYTES + 64 * len(tasks)


class Inflater:
    """Batch DEFLATE decoder on the GPU (include/tredgpu.h section 4): one HIP stream with pinned staging; one per host
    thread.  ``reserve`` hands out numpy views of the staging buffers -- compressed payloads and their offsets are
    written into them, the inflated bytes are read from ``out`` in place after ``run``."""

    def __init__(self, device=0, host_out=True):
        """host_out=False: no pinned host copy of the whole output (run / fetch are then not available: run_walk and
        fetch_dense are) -- 45 MB less page-locked memory per 30x sample of a call."""
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.tredgpu_inflater_create(device, C.byref(self._h))
        if rc != 0:
            raise TredGpuError("tredgpu_inflater_create: %s (rc=%d)" % (self._lib.tredgpu_inflater_last_error(None).decode(), rc))
        self.comp_addr = self.out_addr = 0
        self.host_out = bool(host_out)
        if not host_out:
            self._check(self._lib.tredgpu_inflater_host_out(self._h, 0), "tredgpu_inflater_host_out")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.tredgpu_inflater_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the module globals may be gone
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise TredGpuError("%s: %s (rc=%d)" % (what, self._lib.tredgpu_inflater_last_error(self._h).decode(), rc))
        return rc

    def reserve(self, comp_bytes, out_bytes, n_blocks):
        """(comp, out, comp_off, out_off): uint8 views of comp_bytes / out_bytes bytes, int64 views of n_blocks + 1."""
        ptr = [C.c_void_p() for _ in range(4)]
        self._check(self._lib.tredgpu_inflater_reserve(self._h, comp_bytes, out_bytes, n_blocks, *[C.byref(p) for p in ptr]),
                    "tredgpu_inflater_reserve")
        self.comp_addr, self.out_addr = ptr[0].value, ptr[1].value or 0

        def view(p, ctype, n):
            if not p.value:
                return None                   # (host_out=False: there is no host copy of the output)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(max(n, 1),))[:n]
        return (view(ptr[0], C.c_uint8, comp_bytes), view(ptr[1], C.c_uint8, out_bytes),
                view(ptr[2], C.c_int64, n_blocks + 1), view(ptr[3], C.c_int64, n_blocks + 1))

    def run(self, n_blocks, crc=False):
        """Decodes the n_blocks blocks laid out in the reserved buffers; returns the int32 status per block -- with
        ``crc=True`` (status, uint32 CRC-32 of every block's inflated bytes, computed on the device)."""
        status = np.zeros(max(n_blocks, 1), np.int32)
        if not crc:
            self._check(self._lib.tredgpu_inflate_blocks(self._h, n_blocks, status.ctypes.data), "tredgpu_inflate_blocks")
            return status[:n_blocks]
        sums = np.zeros(max(n_blocks, 1), np.uint32)
        self._check(self._lib.tredgpu_inflate_blocks_crc(self._h, n_blocks, status.ctypes.data, sums.ctypes.data), "tredgpu_inflate_blocks_crc")
        return status[:n_blocks], sums[:n_blocks]

    def run_walk(self, n_blocks, blk_coffset, blk_clen, blk_crc, tasks, chunks, pairs_per_task=2048, alt_tasks=None, alt_chunks=None,
                 pool_pairs=None, select=None):
        """tredgpu_inflate_walk: decodes the blocks laid out in the reserved buffers and walks the pair-length regions
        (tasks WALK_TASK_DTYPE, chunks WALK_CHUNK_DTYPE) over them on the device.  No block is copied back (fetch does
        that).  Returns (status, crc, results WALK_RESULT_DTYPE, global pool, target pool) -- with alt_tasks / alt_chunks
        (the alternative loci's walks) also (results ALT_RESULT_DTYPE, uint8 flags of the blocks that hold their records).
        pool_pairs: room in the global pool for the whole call (walk_pool_pairs: a bound from the planned bytes, so that
        no coverage makes a region fall back to the host for want of room); without it pairs_per_task per task.
        select (SELECT_TASK_DTYPE, one per task): the read selection, depth sums and sizes where the records are (tredgpu.h
        section 5) -- the results (SELECT_RESULT_DTYPE) are appended to the returned tuple, the selected records stay on the
        device for Context.genotype_selected."""
        status, sums = np.zeros(max(n_blocks, 1), np.int32), np.zeros(max(n_blocks, 1), np.uint32)
        coff = np.ascontiguousarray(blk_coffset, np.int64)
        clen = np.ascontiguousarray(blk_clen, np.int32)
        xcrc = np.ascontiguousarray(blk_crc, np.uint32)
        tasks = np.ascontiguousarray(tasks, WALK_TASK_DTYPE)
        chunks = np.ascontiguousarray(chunks, WALK_CHUNK_DTYPE)
        if not (len(coff) == len(clen) == len(xcrc) == n_blocks):
            raise ValueError("one compressed offset / length / CRC per block")
        res = np.zeros(max(len(tasks), 1), WALK_RESULT_DTYPE)
        room = len(tasks) * int(pairs_per_task) if pool_pairs is None else int(pool_pairs)
        gp = np.zeros(room + 4096, np.int32)
        tp = np.zeros(max(room // 8, 16 * len(tasks)) + 1024, np.int32)
        n_alt = 0 if alt_tasks is None else len(alt_tasks)
        at = np.ascontiguousarray(alt_tasks if n_alt else np.zeros(1, WALK_TASK_DTYPE), WALK_TASK_DTYPE)
        ac = np.ascontiguousarray(alt_chunks if (n_alt and len(alt_chunks)) else np.zeros(1, WALK_CHUNK_DTYPE), WALK_CHUNK_DTYPE)
        ares = np.zeros(max(n_alt, 1), ALT_RESULT_DTYPE)
        need = np.zeros(max(n_blocks, 1), np.uint8)
        sel = selres = None
        if select is not None:
            sel = np.ascontiguousarray(select, SELECT_TASK_DTYPE)
            if len(sel) != len(tasks):
                raise ValueError("one select task per walk task")
            selres = np.zeros(max(len(tasks), 1), SELECT_RESULT_DTYPE)
        a = WalkArgs(coff.ctypes.data, clen.ctypes.data, xcrc.ctypes.data, tasks.ctypes.data, len(tasks), chunks.ctypes.data,
                     len(chunks), res.ctypes.data, gp.ctypes.data, len(gp), tp.ctypes.data, len(tp), 0, 0,
                     at.ctypes.data, n_alt, ac.ctypes.data, len(alt_chunks) if n_alt else 0, ares.ctypes.data, need.ctypes.data,
                     sel.ctypes.data if sel is not None and len(sel) else None, selres.ctypes.data if sel is not None and len(sel) else None)
        self._check(self._lib.tredgpu_inflate_walk(self._h, n_blocks, status.ctypes.data, sums.ctypes.data, C.byref(a)),
                    "tredgpu_inflate_walk")
        out = (status[:n_blocks], sums[:n_blocks], res[:len(tasks)], gp[:a.n_global], tp[:a.n_target])
        if alt_tasks is not None:
            out = out + (ares[:n_alt], need[:n_blocks])
        return out if select is None else out + (selres[:len(tasks)],)

    def fetch(self, need):
        """tredgpu_inflater_fetch: the blocks with need[k] != 0 of the last run_walk, to their places in ``out``."""
        need = np.ascontiguousarray(need, np.uint8)
        return self._check(self._lib.tredgpu_inflater_fetch(self._h, len(need), need.ctypes.data), "tredgpu_inflater_fetch")

    def fetch_dense(self, need):
        """tredgpu_inflater_fetch_dense: the blocks with need[k] != 0 of the last run_walk (and short gaps between them),
        one after the other in a pinned buffer of their own; returns (address, int64 offsets[n + 1]): block k lies at
        address + offsets[k] and is offsets[k + 1] - offsets[k] bytes long (0: not copied)."""
        need = np.ascontiguousarray(need, np.uint8)
        off = np.zeros(len(need) + 1, np.int64)
        host = C.c_void_p()
        self._check(self._lib.tredgpu_inflater_fetch_dense(self._h, len(need), need.ctypes.data, C.byref(host), off.ctypes.data),
                    "tredgpu_inflater_fetch_dense")
        return host.value or 0, off

    def pinned_bytes(self):
        """Page-locked host memory this inflater holds, in bytes."""
        return int(self._lib.tredgpu_inflater_pinned_bytes(self._h)) if self._h else 0

    def walk_serial_regions(self):
        """Regions of the last run_walk whose records the serial chain listed (the lane-parallel one handed them back)."""
        return int(self._lib.tredgpu_inflater_walk_serial_regions(self._h))

    def walk_ms(self):
        a = C.c_double()
        self._check(self._lib.tredgpu_inflater_walk_ms(self._h, C.byref(a)), "tredgpu_inflater_walk_ms")
        return a.value

    def timing(self):
        """(total_ms, kernel_ms) of the last call on the device."""
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.tredgpu_inflater_timing(self._h, C.byref(a), C.byref(b)), "tredgpu_inflater_timing")
        return a.value, b.value
