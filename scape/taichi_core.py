"""Import path of the reference's kernel seam (apa_core.py:23), served by HIP kernels."""
from scape_amd.taichi_core import (  # noqa: F401
    loglik_xlr_t_pa, loglik_xlr_t_r_known, loglik_xlr_t_r_unknown, get_loglik_marginal_tensor,
    loglik_marginal_lxr, neg_infinite, pos_infinite, PI)
