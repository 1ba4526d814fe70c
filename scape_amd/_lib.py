"""ctypes binding of libscape_hip.so (include/scape_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is
present, every compute entry point raises :class:`ScapeHipError`.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

SENT = float(np.finfo("f").min)   # reference taichi_core.py:8 / apa_core.py:428
MAX_BETA, MAX_S, MAX_K = 160, 64, 63
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCAPE_HIP_LIB") or os.path.join(_HERE, "libscape_hip.so")   # override: A/B builds only

c_d, c_i, c_i32, c_i64 = ctypes.c_double, ctypes.c_int, ctypes.c_int32, ctypes.c_int64
P_d = ctypes.POINTER(c_d)
P_i32 = ctypes.POINTER(c_i32)
P_i64 = ctypes.POINTER(c_i64)
P_i8 = ctypes.POINTER(ctypes.c_int8)
P_void = ctypes.c_void_p


class ScapeHipError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """struct scape_hip_params"""
    _fields_ = [("mu_f", c_d), ("sigma_f", c_d), ("max_unif_ws", c_d),
                ("n_beta", c_i32), ("n_s", c_i32), ("nround", c_i32), ("reserved", c_i32),
                ("betas", c_d * MAX_BETA), ("s_dis", c_d * MAX_S), ("pmf_s", c_d * MAX_S)]


# name -> (restype, argtypes); must list every symbol include/scape_hip.h declares
SIGNATURES = {
    "scape_hip_abi_version": (c_i, []),
    "scape_hip_last_error": (ctypes.c_char_p, []),
    "scape_hip_device_count": (c_i, [ctypes.POINTER(c_i)]),
    "scape_hip_create": (c_i, [c_i, ctypes.POINTER(P_void)]),
    "scape_hip_destroy": (c_i, [P_void]),
    "scape_hip_device_name": (c_i, [P_void, ctypes.c_char_p, c_i]),
    "scape_hip_loglik_xlr_t_pa": (c_i, [P_void, P_d, P_d, P_d, c_i32, c_d, c_d, P_d]),
    "scape_hip_loglik_xlr_t_r_known": (c_i, [P_void, P_d, P_d, P_d, c_i32, P_d, P_d, c_i32, c_d, c_d, c_d, P_d]),
    "scape_hip_loglik_xlr_t_r_unknown": (c_i, [P_void, P_d, P_d, P_d, c_i32, P_d, P_d, c_i32, c_d, c_d, c_d, P_d]),
    "scape_hip_get_loglik_marginal_tensor": (c_i, [P_void, P_d, c_i32, P_d, c_i32, P_d, c_i32, P_d]),
    "scape_hip_batch_load": (c_i, [P_void, ctypes.POINTER(Params), c_i32, P_i64, P_d, P_d, P_d, P_d, P_d,
                                   P_i64, P_d, P_d, P_d, P_d]),
    "scape_hip_batch_bytes": (c_i, [P_void, P_i64, P_i64, P_i64]),
    "scape_hip_batch_phase_b_form": (c_i, [P_void, P_i32]),
    "scape_hip_batch_build": (c_i, [P_void]),
    "scape_hip_batch_em": (c_i, [P_void, c_i32, c_i32, P_i32, P_i32, P_i32, P_i32, P_i32, P_d, P_i8,
                                 P_i32, P_i32, P_d, P_d, P_i32, P_d]),
    "scape_hip_batch_em_fetch_lb": (c_i, [P_void, c_i32, P_i32, P_d]),
    "scape_hip_batch_labels": (c_i, [P_void, c_i32, c_i32, P_i32, P_i32, P_i32, P_i32, P_d, P_i32]),
    "scape_hip_batch_fetch_loglik": (c_i, [P_void, c_i32, P_d]),
    "scape_hip_batch_fetch_tensor": (c_i, [P_void, c_i32, P_d]),
    "scape_hip_batch_free": (c_i, [P_void]),
    "scape_hip_timing_reset": (c_i, [P_void]),
    "scape_hip_timing_get": (c_i, [P_void, c_i32, P_d, P_i32]),
    "scape_hip_em_counters": (c_i, [P_void, P_i64, P_i64, P_i64]),
    "scape_hip_em_traffic": (c_i, [P_void, P_i64, P_i64, P_i64, P_i64]),
}

_lib = None


def load_library():
    """dlopen libscape_hip.so and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ScapeHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); scape_amd has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.scape_hip_abi_version() != 4:
            raise ScapeHipError("libscape_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def last_error():
    return load_library().scape_hip_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise ScapeHipError(f"{what}: {last_error()}")


def ptr(a, t=P_d):
    return a.ctypes.data_as(t)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_count():
    n = c_i(0)
    check(load_library().scape_hip_device_count(ctypes.byref(n)), "device_count")
    return n.value


class Context:
    """One opaque device context (one per GPU, one host thread per context)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = P_void()
        check(self.lib.scape_hip_create(int(device), ctypes.byref(self.h)), f"create(device={device})")
        self.device = device

    def close(self):
        if self.h:
            self.lib.scape_hip_destroy(self.h)
            self.h = P_void()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def name(self):
        buf = ctypes.create_string_buffer(256)
        check(self.lib.scape_hip_device_name(self.h, buf, 256), "device_name")
        return buf.value.decode()


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("SCAPE_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
