"""scape_amd - MI355X-native implementation of SCAPE's ``infer_pa`` hot path.

Layout: ``csrc/`` HIP kernels + C-ABI (``include/scape_hip.h``); ``_lib`` ctypes binding;
``taichi_core`` the reference's operator seam; ``host`` binning/grids/restart sampling;
``engine`` batched driver + model selection; ``apa_core`` CLI / Parameters; ``dist``
multi-GPU sharding; ``synth`` synthetic pile-ups; ``safe_pickle`` non-executing chunk reader.
"""
__version__ = "0.1.0"
