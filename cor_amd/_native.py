"""ctypes binding of libcor_amd.so (C ABI declared in include/cor_amd.h).

There is NO fallback: if the shared library is missing or a call returns non-zero, this raises.
Build with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C cor_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COR_AMD_LIB") or os.path.join(_HERE, "csrc", "libcor_amd.so")   # COR_AMD_LIB: another build of the SAME ABI (same-box A/B runs of tools/)

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_GELU_ERF, ACT_RELU, ACT_SIGMOID, ACT_GELU_TANH = 0, 1, 2, 3, 4
EINVAL, ENOSUPPORT = -1, -2
TOPK_FORCE_LISTS, TOPK_NO_FALLBACK, TOPK_FORCE_GLOBAL_THRESHOLD, TOPK_WAVE_FINAL = 1, 2, 8, 16
ORDER_REVERSE = 1 << 30    # cor_gemm cfg / cor_layernorm act / cor_sam_attention variant: walk the work from the last item to the first
KERNEL_ROWLANE, KERNEL_FEWQ, KERNEL_FLASH_MFMA, KERNEL_FLASH_PIPELINED, KERNEL_WINDOW_BLOCK = 1, 2, 3, 4, 5
import numpy as _np
Q_PRESCALE_HD64 = float(_np.float32(0.125) * _np.float32(1.4426950408889634))   # scale * log2(e) for head_dim 64, as a float32

_p, _i, _l, _f, _ll = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_longlong

# name -> argtypes (restype int unless noted); order follows include/cor_amd.h exactly
SIGNATURES = {
    "cor_version": [],
    "cor_gemm": [_p, _l, _p, _l, _i, _p, _l, _i, _i, _i, _i, _p, _i, _p, _p, _l, _i, _i, _p],
    "cor_layernorm": [_p, _i, _p, _i, _p, _p, _i, _i, _f, _i, _p],
    "cor_attention": [_p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _p, _l, _l, _i, _i, _i, _i, _i, _i, _f, _p],
    "cor_attention_kernel_id": [_i, _i, _i, _i, _i, _i],
    "cor_sam_attention": [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "cor_patchify": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "cor_im2col3x3": [_p, _i, _p, _i, _i, _i, _i, _p],
    "cor_add": [_p, _i, _p, _i, _p, _i, _l, _l, _p],
    "cor_copy_rows": [_p, _l, _i, _p, _l, _i, _i, _i, _p],
    "cor_tokens_to_nchw": [_p, _i, _p, _i, _i, _i, _p],
    "cor_nchw_to_tokens": [_p, _p, _i, _i, _i, _i, _p],
    "cor_l2norm_rows": [_p, _i, _p, _i, _i, _i, _f, _p],
    "cor_embed_tokens": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cor_bilinear": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "cor_conv3x3s2_small": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "cor_dwconv7x7": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "cor_adapter_pool": [_p, _p, _p, _i, _i, _i, _i, _p],
    "cor_masked_pool": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "cor_fuse_gate": [_p, _p, _p, _p, _p, _i, _i, _p],
    "cor_fuse_mix": [_p, _p, _p, _i, _i, _p],
    "cor_dense_pe": [_p, _p, _i, _i, _p],
    "cor_upscale_shuffle": [_p, _i, _p, _p, _p, _f, _i, _p, _i, _i, _i, _i, _i, _p],
    "cor_upscale_hyper": [_p, _i, _p, _p, _p, _l, _p, _i, _i, _i, _i, _i, _i, _p],
    "cor_iou_select": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "cor_decoder_heads": [_p, _p, _p, _p, _p, _i, _p, _p, _i, _p],
    "cor_mask_prob_minmax": [_p, _p, _i, _i, _p],
    "cor_resize_binarize": [_p, _p, _i, _i, _i, _i, _i, _f, _p],
    "cor_resize_gray": [_p, _p, _i, _i, _i, _i, _i, _p],
    "cor_resample_rows_u8": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "cor_resample_cols_u8": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "cor_mask_metrics": [_p, _p, _p, _i, _i, _f, _p],
    "cor_topk_workspace_bytes": [_i, _i, _i],
    "cor_similarity_topk": [_p, _p, _i, _i, _i, _i, _i, _ll, _p, _p, _p, _i, _p],
}
_RESTYPE = {"cor_topk_workspace_bytes": _l}

_lib = None


class NativeError(RuntimeError):
    pass


def use_probe_library():
    """Development tools only (tools/attn_stamps.py, tools/gemm_ksweep.py): bind the COR_PROBES build (make -C cor_amd/csrc probes ->
    tools/probes/libcor_probes.so: timing probes compiled in) instead of the product library. Must be called before load()."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise RuntimeError("use_probe_library() must be called before the library is first loaded")
    LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "probes", "libcor_probes.so")


def load():
    """Load (once) and return the ctypes library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library were resolved first it would bind the
    # system copy under /opt/rocm and the kernels would launch into a runtime torch never initialised (hipErrorNoDevice at the first
    # launch: seen when __graft_entry__.build() and smoke() ran in one process). Importing torch first makes both use torch's copy.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"cor_amd: HIP extension not built ({LIB_PATH} missing). There is no CPU fallback. "
            "Build it with: python -c 'import __graft_entry__ as g; g.build()'  (or make -C cor_amd/csrc)")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _i)
    _lib = lib
    return lib


def check(rc: int, name: str):
    if rc == 0:
        return
    if rc == EINVAL:
        raise NativeError(f"{name}: invalid argument")
    if rc == ENOSUPPORT:
        raise NativeError(f"{name}: no kernel for this shape/dtype (no fallback by design)")
    raise NativeError(f"{name}: HIP error {rc}")
