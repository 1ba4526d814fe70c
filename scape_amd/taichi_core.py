"""Drop-in for the reference's L3 seam ``scape/taichi_core.py`` on MI355X.

``apa_core.py:23`` of the reference imports exactly these four host functions; the
names, argument order and array conventions (contiguous f64 numpy in, new f64 numpy
out) are the reference's (``taichi_core.py:183``, ``:200``, ``:210``, ``:237``).
Each call goes through the ctypes C-ABI to a hand-written HIP kernel; there is no
CPU path.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check, default_context, f64, ptr

pos_infinite = np.finfo("f").max
neg_infinite = np.finfo("f").min
PI = 3.141592653589793


def loglik_xlr_t_pa(x_arr, l_arr, pa_arr, theta, sigma_f):
    """reference taichi_core.py:183-197"""
    ctx = default_context()
    x, l, pa = f64(x_arr), f64(l_arr), f64(pa_arr)
    if not (len(x) == len(l) == len(pa)):
        raise ValueError("x_arr, l_arr, pa_arr must have the same length")
    out = np.zeros(len(x))
    check(ctx.lib.scape_hip_loglik_xlr_t_pa(ctx.h, ptr(x), ptr(l), ptr(pa), len(x), float(theta),
                                            float(sigma_f), ptr(out)), "loglik_xlr_t_pa")
    return out


def loglik_xlr_t_r_known(x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr, theta, mu_f, sigma_f):
    """reference taichi_core.py:200-207"""
    ctx = default_context()
    x, l, r, s, pmf = map(f64, (x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr))
    if not (len(x) == len(l) == len(r)) or len(s) != len(pmf):
        raise ValueError("array lengths do not match")
    out = np.zeros(len(x))
    check(ctx.lib.scape_hip_loglik_xlr_t_r_known(ctx.h, ptr(x), ptr(l), ptr(r), len(x), ptr(s), ptr(pmf),
                                                 len(s), float(theta), float(mu_f), float(sigma_f),
                                                 ptr(out)), "loglik_xlr_t_r_known")
    return out


def loglik_xlr_t_r_unknown(x_arr, l_arr, r_arr, s_dis_arr, pmf_s_arr, theta, mu_f, sigma_f):
    """reference taichi_core.py:210-215 (``r_arr`` is unused there too)"""
    ctx = default_context()
    x, l, s, pmf = map(f64, (x_arr, l_arr, s_dis_arr, pmf_s_arr))
    if len(x) != len(l) or len(s) != len(pmf):
        raise ValueError("array lengths do not match")
    out = np.zeros(len(x))
    check(ctx.lib.scape_hip_loglik_xlr_t_r_unknown(ctx.h, ptr(x), ptr(l), None, len(x), ptr(s), ptr(pmf),
                                                   len(s), float(theta), float(mu_f), float(sigma_f),
                                                   ptr(out)), "loglik_xlr_t_r_unknown")
    return out


def get_loglik_marginal_tensor(all_theta, predef_beta_arr, loglik_xlr_t_arr):
    """reference taichi_core.py:237-246; returns [n_theta, n_beta, n_frag]"""
    ctx = default_context()
    th, be, A = f64(all_theta), f64(predef_beta_arr), f64(loglik_xlr_t_arr)
    if A.ndim != 2 or A.shape[1] != len(th):
        raise ValueError("loglik_xlr_t_arr must be [n_frag, len(all_theta)]")
    out = np.zeros((len(th), len(be), A.shape[0]))
    check(ctx.lib.scape_hip_get_loglik_marginal_tensor(ctx.h, ptr(th), len(th), ptr(be), len(be), ptr(A),
                                                       A.shape[0], ptr(out)), "get_loglik_marginal_tensor")
    return out


def loglik_marginal_lxr(alpha, beta, all_theta, loglik_xlr_t_arr):
    """reference taichi_core.py:218-234 (one (alpha, beta) slice of the tensor)"""
    th = f64(all_theta)
    A = f64(loglik_xlr_t_arr)
    i = int(np.searchsorted(th, alpha, side="left"))
    if i >= len(th) or th[i] != alpha:
        raise ValueError("alpha must be a point of all_theta")
    hi = int(np.searchsorted(th, alpha + 3 * beta, side="right"))      # window end (exclusive)
    if hi > A.shape[1]:
        raise IndexError("loglik_xlr_t_arr has fewer columns than the +-3 beta window needs")
    # the reference only touches the window's columns, so a narrower matrix is accepted (:177-178)
    return get_loglik_marginal_tensor(th[:A.shape[1]], np.array([beta], dtype=np.float64), A)[i, 0]
