#!/usr/bin/env python3
"""Import the Python-2 reference (humanlongevity/tredparse) under Python 3, in memory.

BUILD-CONTAINER ONLY: reads /root/reference at run time; no reference text is stored in this repo
and nothing here travels to the GPU box.  Used by tools/gen_golden.py to produce the committed
golden vectors under tests/golden/.

How: each reference module's source is run through lib2to3 (print statements, xrange, dict
iterators, ...), then an AST pass rewrites every `a / b` into a helper that floor-divides when both
operands are integers (Python-2 semantics, which the reference relies on, e.g. models.py:156,159,
bam_parser.py:133) and `x.next()` into `next(x)`; the result is exec'd into synthetic modules
registered under both their package name and the bare name the reference imports implicitly.
pysam (absent here) is replaced by a stand-in object supplied by the caller.
"""
import ast
import builtins
import numbers
import os
import sys
import types
import warnings

REF = os.environ.get("TRED_REFERENCE", "/root/reference")


def _py2div(a, b):
    if isinstance(a, numbers.Integral) and isinstance(b, numbers.Integral):
        return a // b
    return a / b


class _Py2(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[]),
                node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if isinstance(f, ast.Attribute) and f.attr == "next" and not node.args:
            return ast.copy_location(
                ast.Call(func=ast.Name(id="next", ctx=ast.Load()), args=[f.value], keywords=[]), node)
        return node


def _to_py3(src, name):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        tool = refactor.RefactoringTool(fixers)
        out = str(tool.refactor_string(src + "\n", name))
    out = out.replace("string.maketrans", "str.maketrans")
    tree = _Py2().visit(ast.parse(out, name))
    ast.fix_missing_locations(tree)
    return compile(tree, name, "exec")


def _load(relpath, modname, aliases=(), extra=None):
    path = os.path.join(REF, relpath)
    with open(path) as fp:
        src = fp.read()
    mod = types.ModuleType(modname)
    mod.__file__ = path
    if "." in modname:
        mod.__package__ = modname.rsplit(".", 1)[0]
    mod.__dict__["_py2div"] = _py2div
    mod.__dict__["xrange"] = builtins.range
    mod.__dict__["range"] = lambda *a: list(builtins.range(*a))
    if extra:
        mod.__dict__.update(extra)
    sys.modules[modname] = mod
    for a in aliases:
        sys.modules[a] = mod
    exec(_to_py3(src, path), mod.__dict__)
    return mod


def load_reference(pysam_standin=None, libssw_dir=None, full=False):
    """Returns a namespace with .ssw (ssw_wrap), .utils, .bam_parser, .models of the reference
    (+ .meta and .tred with full=True)."""
    libssw_dir = libssw_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref")
    # ssw_wrap.load_ssw_library falls back to the bare soname (ssw_wrap.py:24-31): preload it
    import ctypes
    ctypes.CDLL(os.path.join(libssw_dir, "libssw.so"), mode=ctypes.RTLD_GLOBAL)
    os.environ["LD_LIBRARY_PATH"] = libssw_dir + ":" + os.environ.get("LD_LIBRARY_PATH", "")
    real_cdll_load = ctypes.cdll.LoadLibrary

    def _loader(name):
        if os.path.basename(name) == "libssw.so":
            return real_cdll_load(os.path.join(libssw_dir, "libssw.so"))
        return real_cdll_load(name)

    ctypes.cdll.LoadLibrary = _loader
    try:
        pkg = types.ModuleType("tredparse")
        pkg.__path__ = []
        pkg.__version__ = "0.7.8"
        sys.modules["tredparse"] = pkg
        sys.modules["pysam"] = pysam_standin if pysam_standin is not None else types.ModuleType("pysam")
        ssw = _load("src/ssw_wrap.py", "ssw", aliases=("ssw_wrap",))
        utils = _load("tredparse/utils.py", "tredparse.utils", aliases=("utils",))
        bam_parser = _load("tredparse/bam_parser.py", "tredparse.bam_parser", aliases=("bam_parser",))
        models = _load("tredparse/models.py", "tredparse.models", aliases=("models",))
    finally:
        ctypes.cdll.LoadLibrary = real_cdll_load
    ns = types.SimpleNamespace(ssw=ssw, utils=utils, bam_parser=bam_parser, models=models)
    if full:
        import pandas as pd
        real_read_csv = pd.read_csv

        def read_csv_latin1(*a, **k):   # TREDs.meta.csv is not valid UTF-8 (py2 never decoded it)
            k.setdefault("encoding", "latin-1")
            return real_read_csv(*a, **k)
        pd.read_csv = read_csv_latin1
        ns.meta = _load("tredparse/meta.py", "tredparse.meta", aliases=("meta",))
        ns.tred = _load("tredparse/tred.py", "tredparse.tred")
    return ns


if __name__ == "__main__":
    ref = load_reference()
    a = ref.ssw.Aligner(ref_seq="ACGTACGTTTGACCA" * 4, match=1, mismatch=5, gap_open=7, gap_extend=2)
    r = a.align("ACGTTTGACCAACGTACG" * 2, min_score=10, min_len=5)
    print(r.score, r.ref_begin, r.ref_end, r.query_begin, r.query_end)
    print(ref.models.SMALL_VALUE, ref.bam_parser.FLANKMATCH)
