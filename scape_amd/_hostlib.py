"""ctypes binding of libscape_host.so (include/scape_host.h): restart sampling in native code.

The numbers are the ones numpy's legacy RandomState would produce in the reference's call order
(apa_core.py:653-677, :781-829); scape_amd/host.py::Sampler is the readable twin and the
fallback for inputs the C path declines (status 1).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscape_host.so")
MAX_K = 64
STATE_WORDS = 625

c_d, c_i, c_i32, c_u32 = ctypes.c_double, ctypes.c_int, ctypes.c_int32, ctypes.c_uint32
P_d = ctypes.POINTER(c_d)
P_i8 = ctypes.POINTER(ctypes.c_int8)
P_i32 = ctypes.POINTER(c_i32)
P_i64 = ctypes.POINTER(ctypes.c_int64)
P_u32 = ctypes.POINTER(c_u32)


class PlanArgs(ctypes.Structure):
    """struct scape_host_plan_args"""
    _fields_ = [("n_utr", c_i32), ("n_threads", c_i32), ("n_trial", c_i32), ("n_round", c_i32),
                ("kmax", c_i32), ("kcap", c_i32),
                ("seeds", P_u32), ("peak_off", P_i64), ("peaks", P_d), ("peak_w", P_d),
                ("theta_off", P_i64), ("theta", P_d),
                ("L", P_i32), ("n_max", P_i32), ("n_min", P_i32), ("n_beta", P_i32),
                ("shift_scale", P_d), ("max_unif_ws", P_d), ("spans", P_i64),
                ("ju", P_i32), ("jk", P_i32), ("a", P_i32), ("b", P_i32), ("w", P_d), ("ka", P_i8),
                ("states", P_u32), ("prune_w", P_d), ("prune_ka", P_i8), ("prune_states", P_u32),
                ("status", P_i32)]


class SweepItem(ctypes.Structure):
    """struct scape_host_sweep_item"""
    _fields_ = [("state625", P_u32), ("peaks", P_d), ("peak_w", P_d), ("theta", P_d),
                ("n_peak", c_i32), ("T", c_i32), ("L", c_i32), ("n_beta", c_i32),
                ("shift_scale", c_d), ("max_unif_ws", c_d),
                ("n_max", c_i32), ("n_min", c_i32), ("kmax", c_i32), ("status", c_i32),
                ("jk", P_i32), ("a", P_i32), ("b", P_i32), ("w", P_d), ("ka", P_i8)]


# name -> (restype, argtypes); must list every symbol include/scape_host.h declares
SIGNATURES = {
    "scape_host_abi_version": (c_i, []),
    "scape_host_mt_seed": (None, [c_u32, P_u32]),
    "scape_host_mt_double": (c_d, [P_u32]),
    "scape_host_init_ws": (c_i, [P_u32, c_i, c_d, P_d]),
    "scape_host_k_arr": (c_i, [P_u32, c_i, c_i, P_i8]),
    "scape_host_init_job": (c_i, [P_u32, P_d, P_d, c_i, P_d, c_i, c_i, c_i, c_d, c_d, c_i, c_i,
                                  P_i32, P_i32, P_d, P_i8]),
    "scape_host_sweep": (c_i, [P_u32, P_d, P_d, c_i, P_d, c_i, c_i, c_i, c_d, c_d, c_i, c_i, c_i, c_i, c_i,
                               P_i32, P_i32, P_i32, P_d, P_i8]),
    "scape_host_sweep_batch": (c_i, [ctypes.POINTER(SweepItem), c_i, c_i, c_i, c_i]),
    "scape_host_plan": (c_i, [ctypes.POINTER(PlanArgs)]),
}

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.scape_host_abi_version() != 2:
            raise RuntimeError("libscape_host.so ABI version mismatch")
        _lib = lib
    return _lib


def ptr(a, t):
    return a.ctypes.data_as(t)


def state_from_numpy(st):
    """RandomState.get_state() tuple -> uint32[625]"""
    out = np.empty(STATE_WORDS, dtype=np.uint32)
    out[:624] = st[1]
    out[624] = st[2]
    return out


def state_to_numpy(words, has_gauss=0, cached=0.0):
    return ("MT19937", np.array(words[:624], dtype=np.uint32), int(words[624]), int(has_gauss), float(cached))


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))
