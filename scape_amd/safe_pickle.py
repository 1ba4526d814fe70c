"""Non-executing reader for the pickle streams on the ``infer_pa`` boundary.

``scape prepare_input`` writes each chunk as a concatenation of pickles, one
``(gene_info_str, DataFrame)`` tuple per UTR (reference
``src/scape/input_processor.py:223-259``), and ``infer_pa`` writes one
``scape.apa_core.Parameters`` per UTR (``src/scape/apa_core.py:1134-1137``).
``pickle.load`` would import and call whatever the file names.  This module
instead decodes the opcode stream with :func:`pickletools.genops` (a pure
decoder) and runs a tiny stack machine that only ever builds *symbolic* nodes
for GLOBAL / REDUCE / NEWOBJ / BUILD.  Nothing named by the file is imported
or called; the handful of data types that occur on this boundary (numpy
arrays/scalars/dtypes, the pandas 1.x/2.x ``BlockManager`` layout, ``slice``,
``Parameters``) are rebuilt by the code below from the raw bytes.
"""
from __future__ import annotations

import io
import pickletools
import threading
from types import SimpleNamespace

import numpy as np


class UnsafePickleError(ValueError):
    """The stream uses an opcode or a global this reader does not model."""


class Global:
    __slots__ = ("module", "name")

    def __init__(self, module, name):
        self.module, self.name = module, name

    @property
    def path(self):
        return f"{self.module}.{self.name}"

    def __repr__(self):
        return f"<global {self.path}>"


class Node:
    """Symbolic result of REDUCE / NEWOBJ (+ BUILD state, list/dict items)."""
    __slots__ = ("func", "args", "state", "items", "kv")

    def __init__(self, func, args):
        self.func, self.args = func, args
        self.state = None
        self.items = []     # APPEND(S) on a non-list
        self.kv = []        # SETITEM(S) on a non-dict

    def __repr__(self):
        return f"<node {self.func!r} args={len(self.args) if isinstance(self.args, tuple) else '?'}>"


_MARK = object()


def _pop_mark(stack):
    items = []
    while True:
        v = stack.pop()
        if v is _MARK:
            break
        items.append(v)
    items.reverse()
    return items


def _run(ops):
    """Execute one pickle's opcodes symbolically; returns the top of stack at STOP."""
    stack, memo = [], {}
    for op, arg, _pos in ops:
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            return stack.pop()
        if n == "MARK":
            stack.append(_MARK)
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE", "BINSTRING",
                   "SHORT_BINSTRING", "STRING", "SHORT_BINBYTES", "BINBYTES", "BINBYTES8",
                   "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG", "BINFLOAT",
                   "FLOAT"):
            stack.append(arg)
        elif n == "BYTEARRAY8":
            stack.append(bytearray(arg))
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n == "GLOBAL":
            mod, name = arg.split(" ", 1)
            stack.append(Global(mod, name))
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            mod = stack.pop()
            stack.append(Global(mod, name))
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "TUPLE":
            stack.append(tuple(_pop_mark(stack)))
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "LIST":
            stack.append(list(_pop_mark(stack)))
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "DICT":
            it = _pop_mark(stack)
            stack.append({_hashable(it[i]): it[i + 1] for i in range(0, len(it), 2)})
        elif n == "EMPTY_SET":
            stack.append(set())
        elif n == "FROZENSET":
            stack.append(frozenset(_pop_mark(stack)))
        elif n == "ADDITEMS":
            it = _pop_mark(stack)
            stack[-1].update(it)
        elif n == "APPEND":
            v = stack.pop()
            tgt = stack[-1]
            (tgt if isinstance(tgt, list) else tgt.items).append(v)
        elif n == "APPENDS":
            it = _pop_mark(stack)
            tgt = stack[-1]
            (tgt if isinstance(tgt, list) else tgt.items).extend(it)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            tgt = stack[-1]
            if isinstance(tgt, dict):
                tgt[_hashable(k)] = v
            else:
                tgt.kv.append((k, v))
        elif n == "SETITEMS":
            it = _pop_mark(stack)
            tgt = stack[-1]
            for i in range(0, len(it), 2):
                if isinstance(tgt, dict):
                    tgt[_hashable(it[i])] = it[i + 1]
                else:
                    tgt.kv.append((it[i], it[i + 1]))
        elif n == "REDUCE":
            args = stack.pop()
            func = stack.pop()
            stack.append(Node(func, args))
        elif n == "NEWOBJ":
            args = stack.pop()
            cls = stack.pop()
            stack.append(Node(cls, args))
        elif n == "NEWOBJ_EX":
            kwargs = stack.pop()
            args = stack.pop()
            cls = stack.pop()
            stack.append(Node(cls, (args, kwargs)))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, Node):
                raise UnsafePickleError("BUILD on a non-object")
            tgt.state = state
        elif n == "POP":
            stack.pop()
        elif n == "POP_MARK":
            _pop_mark(stack)
        elif n == "DUP":
            stack.append(stack[-1])
        else:
            raise UnsafePickleError(f"opcode {n} is not modelled by the safe reader")
    raise UnsafePickleError("pickle stream ended without STOP")


def _hashable(k):
    if isinstance(k, (Node, Global)):
        raise UnsafePickleError("object used as a dict key")
    return k


# --------------------------------------------------------------------------
# materialisers: rebuild data objects from symbolic nodes with OUR code
# --------------------------------------------------------------------------
_NP_RECONSTRUCT = {"numpy.core.multiarray._reconstruct", "numpy._core.multiarray._reconstruct"}
_NP_SCALAR = {"numpy.core.multiarray.scalar", "numpy._core.multiarray.scalar"}
_NP_DTYPE = {"numpy.dtype"}
_PD_UNPICKLE_BLOCK = {"pandas._libs.internals._unpickle_block"}
_PD_NEW_BLOCK = {"pandas.core.internals.blocks.new_block"}
_PD_BLOCKMANAGER = {"pandas.core.internals.managers.BlockManager"}
_PD_DATAFRAME = {"pandas.core.frame.DataFrame"}
_PD_NEW_INDEX = {"pandas.core.indexes.base._new_Index"}
_PARAMETERS = {"scape.apa_core.Parameters", "scape_amd.apa_core.Parameters"}


def _gpath(x):
    return x.path if isinstance(x, Global) else None


def _dtype(node):
    if isinstance(node, Global):          # e.g. numpy.float64 class used as dtype
        if node.module == "numpy":
            return np.dtype(node.name)
        raise UnsafePickleError(f"unexpected dtype global {node!r}")
    if not (isinstance(node, Node) and _gpath(node.func) in _NP_DTYPE):
        raise UnsafePickleError(f"unexpected dtype node {node!r}")
    code = node.args[0]
    if not isinstance(code, str) or code[0] not in "biufOUSV?":
        raise UnsafePickleError(f"unsupported dtype code {code!r}")
    if code[0] in "VO" and code != "O8" and code != "O":
        raise UnsafePickleError(f"unsupported dtype code {code!r}")
    dt = np.dtype("O" if code[0] == "O" else code)
    if node.state is not None and len(node.state) >= 2 and node.state[1] in ("<", ">"):
        dt = dt.newbyteorder(node.state[1])
    return dt


def _ndarray(node):
    st = node.state
    if st is None or len(st) < 5:
        raise UnsafePickleError("ndarray without state")
    _ver, shape, dt_node, fortran, raw = st[:5]
    dt = _dtype(dt_node)
    shape = tuple(int(s) for s in shape)
    if dt == np.dtype("O"):
        flat = [materialize(v) for v in raw]
        out = np.empty(len(flat), dtype=object)
        for i, v in enumerate(flat):
            out[i] = v
        return out.reshape(shape, order="F" if fortran else "C")
    arr = np.frombuffer(bytes(raw), dtype=dt)
    return arr.reshape(shape, order="F" if fortran else "C").copy()


def _index(node):
    """pandas Index -> numpy array of labels."""
    if isinstance(node, Node) and _gpath(node.func) in _PD_NEW_INDEX:
        cls, d = node.args
        cname = _gpath(cls)
        if cname.endswith("RangeIndex"):
            return np.arange(int(d["start"]), int(d["stop"]), int(d["step"]))
        return np.asarray(materialize(d["data"]))
    raise UnsafePickleError(f"unexpected Index node {node!r}")


def _placement(p):
    if isinstance(p, Node) and _gpath(p.func) == "builtins.slice":
        a, b, c = p.args
        return slice(a, b, c)
    return np.asarray(materialize(p)).astype(np.int64)


class Columns(dict):
    """Column name -> 1-D numpy array: what the infer_pa path needs from a chunk's DataFrame (column access only:
    host.bin_utr, binned.write_binned), without importing pandas (0.3-0.4 s of a 100-UTR `scape infer_pa` run and a
    DataFrame construction per UTR).  ``iter_pickles(path, frames="columns")``."""

    @property
    def columns(self):
        return list(self.keys())

    @property
    def n_rows(self):
        return len(next(iter(self.values()))) if dict.__len__(self) else 0


_tls = threading.local()      # .frames: what _dataframe builds - "pandas" (a DataFrame, default) or "columns" (a Columns mapping)


def _dataframe(node):
    st = node.state
    if not isinstance(st, dict) or "_mgr" not in st:
        raise UnsafePickleError("DataFrame without a _mgr state")
    mgr = st["_mgr"]
    if not (isinstance(mgr, Node) and _gpath(mgr.func) in _PD_BLOCKMANAGER):
        raise UnsafePickleError(f"unexpected manager {mgr!r}")
    blocks, axes = mgr.args[0], mgr.args[1]
    columns = _index(axes[0])
    n_rows = len(_index(axes[1]))
    cols = [None] * len(columns)
    for blk in blocks:
        if not (isinstance(blk, Node) and _gpath(blk.func) in (_PD_UNPICKLE_BLOCK | _PD_NEW_BLOCK)):
            raise UnsafePickleError(f"unexpected block {blk!r}")
        values = np.asarray(materialize(blk.args[0]))
        place = _placement(blk.args[1])
        where = np.arange(len(columns))[place] if isinstance(place, slice) else place
        if values.ndim == 1:
            values = values[None, :]
        for row, ci in enumerate(where):
            cols[int(ci)] = np.ascontiguousarray(values[row])
    if any(c is None or len(c) != n_rows for c in cols):
        raise UnsafePickleError("inconsistent BlockManager layout")
    if getattr(_tls, "frames", "pandas") == "columns":
        return Columns((str(name), col) for name, col in zip(columns, cols))
    import pandas as pd
    return pd.DataFrame({str(name): col for name, col in zip(columns, cols)})


def materialize(obj):
    """Symbolic tree -> plain data (numpy / pandas / containers / SimpleNamespace)."""
    if isinstance(obj, (str, bytes, int, float, bool, type(None), bytearray)):
        return obj
    if isinstance(obj, tuple):
        return tuple(materialize(v) for v in obj)
    if isinstance(obj, list):
        return [materialize(v) for v in obj]
    if isinstance(obj, dict):
        return {k: materialize(v) for k, v in obj.items()}
    if isinstance(obj, (set, frozenset)):
        return type(obj)(materialize(v) for v in obj)
    if isinstance(obj, Global):
        raise UnsafePickleError(f"bare global {obj!r} in data position")
    if isinstance(obj, Node):
        p = _gpath(obj.func)
        if p in _NP_RECONSTRUCT:
            return _ndarray(obj)
        if p in _NP_SCALAR:
            dt, raw = obj.args
            return np.frombuffer(bytes(raw), dtype=_dtype(dt))[0]
        if p in _NP_DTYPE:
            return _dtype(obj)
        if p in _PD_DATAFRAME:
            return _dataframe(obj)
        if p in _PARAMETERS:
            st = obj.state if isinstance(obj.state, dict) else {}
            return SimpleNamespace(**{k: materialize(v) for k, v in st.items()})
        if p == "builtins.slice":
            return slice(*obj.args)
        raise UnsafePickleError(f"global {p} is not on the infer_pa boundary allow-list")
    raise UnsafePickleError(f"unexpected object {type(obj)}")


def iter_pickles(path_or_bytes, frames="pandas"):
    """Yield every top-level object of a concatenated pickle file, materialised.  frames = "columns": DataFrames come
    back as Columns mappings (name -> array) and pandas is never imported."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            data = fh.read()
    bio = io.BytesIO(data)
    n = len(data)
    while bio.tell() < n:
        tree = _run(pickletools.genops(bio))
        keep, _tls.frames = getattr(_tls, "frames", "pandas"), frames      # per thread, only while this object is materialised
        try:
            obj = materialize(tree)
        finally:
            _tls.frames = keep
        yield obj


def load_all(path_or_bytes):
    return list(iter_pickles(path_or_bytes))
