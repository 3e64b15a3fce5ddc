"""ctypes binding of libgcgcn_hip.so (the C ABI declared in include/gcgcn.h).

There is no CPU or eager-PyTorch fallback: if the shared library is missing this module
raises, and every op raises when handed a non-GPU tensor.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCGCN_LIB=<path> loads another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("GCGCN_LIB") or os.path.join(_HERE, "lib", "libgcgcn_hip.so")

ABI_VERSION = 7

SALT_GAT = 0x47415431
SALT_MHA = 0x4D484131
SALT_GCN = 0x47434E31
SALT_GLUE = 0x474C5531

P = c_void_p  # every device pointer crosses the ABI as void*
I = c_int
F = c_float
L = c_int64

class MhaHook(ctypes.Structure):
    """gcgcn_mha_hook: MultiHeadAttention's work riding inside the MultiGraphConvolution calls of the same hop (include/gcgcn.h)."""
    _fields_ = [("flat_q", c_void_p), ("Q", c_void_p), ("P", c_void_p), ("A", c_void_p), ("dQ", c_void_p), ("rng_snap", c_void_p),
                ("p", ctypes.c_float)]


class EdgeRide(ctypes.Structure):
    """gcgcn_edge_ride: an edge-tensor pass riding along with a gcn_fwd / gcn_bwd call (include/gcgcn.h)."""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("D", ctypes.c_int32),
                ("inp", c_void_p), ("n_valid", c_void_p), ("out", c_void_p)]


# name -> (restype, argtypes); kept in the order of include/gcgcn.h
SIGNATURES = {
    "gcgcn_version": (I, []),
    "gcgcn_last_error": (c_char_p, []),
    "gcgcn_set_option": (I, [c_char_p, I]),
    "gcgcn_debug_spread": (I, [L, L, L, L, P, P]),
    "gcgcn_prof_start": (I, [c_char_p, I]),
    "gcgcn_prof_enable": (I, [I]),
    "gcgcn_prof_stop": (I, [P, P, P]),
    "gcgcn_rng_next": (I, [P, P, I, P]),
    "gcgcn_dropout_keep": (I, [P, L, P, c_uint64, F, P]),
    "gcgcn_dropout": (I, [P, P, L, P, c_uint64, F, P]),
    "gcgcn_gat_layout": (I, [I, I, P]),
    "gcgcn_gat_fwd": (I, [I, I, I, I, P, P, P, P, P, F, P, P, P, P, P, P, P, I, P, I, P]),
    "gcgcn_gat_bwd_scratch": (L, [I, I, I]),
    "gcgcn_gat_bwd": (I, [I, I, I, I, P, P, P, P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_edge_mean_fwd": (I, [I, I, I, P, P, P, P]),
    "gcgcn_edge_mean_bwd": (I, [I, I, I, P, P, P, P]),
    "gcgcn_mha_layout": (I, [I, P]),
    "gcgcn_mha_scratch": (L, [I, I, I]),
    "gcgcn_row_blocks_ints": (L, [I, I]),
    "gcgcn_row_blocks": (I, [I, I, P, P, P]),
    "gcgcn_mha_fwd": (I, [I, I, I, I, P, P, P, P, F, P, P, P, P, P, P]),
    "gcgcn_mha_bwd": (I, [I, I, I, I, P, P, P, F, P, P, P, P, P, P, P, P, P, P, I, P, P]),
    "gcgcn_maggc_fusable": (I, [I, I, I]),
    "gcgcn_gcn_layout": (I, [I, I, I, P]),
    "gcgcn_gcn_scratch": (L, [I, I, I, I]),
    "gcgcn_gcn_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, F, P, F, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_gcn_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, F, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_defer_create": (P, []),
    "gcgcn_defer_destroy": (None, [P]),
    "gcgcn_defer_count": (I, [P]),
    "gcgcn_defer_flush": (I, [P, P]),
    "gcgcn_graphconv_fwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_graphconv_bwd": (I, [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_pair_bce_fwd": (I, [I, I, I, P, P, P, P, P, P]),
    "gcgcn_pair_bce_bwd": (I, [I, I, I, P, P, P, P, P, P]),
    "gcgcn_producer_layout": (I, [I, I, P]),
    "gcgcn_producer_count": (I, [I, I, I, I, P, P, P, P]),
    "gcgcn_producer_sizes": (I, [I, I, I, I, I, I, I, L, L, P]),
    "gcgcn_producer_fwd": (I, [I, I, I, I, I, I, I, P, P, P, P, I, P, P, P, P, L, L, P, P, P, L, P, P]),
    "gcgcn_producer_bwd": (I, [I, I, I, I, I, I, I, P, P, P, P, I, P, P, P, P, L, L, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_edge_mean_fwd_compact": (I, [I, I, I, P, P, P, P, P, P]),
    "gcgcn_edge_mean_bwd_compact": (I, [I, I, I, P, P, P, P, P, P, P]),
    "gcgcn_gat_fwd_compact": (I, [I, I, I, I, P, P, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, P]),
    "gcgcn_gat_bwd_compact_scratch": (L, [I, I, I]),
    "gcgcn_gat_bwd_compact": (I, [I, I, I, I, P, P, P, P, P, P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_head_layout": (I, [I, I, I, I, I, P]),
    "gcgcn_head_sizes": (I, [I, I, I, I, P]),
    "gcgcn_head_fwd": (I, [I, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_head_bwd": (I, [I, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_tensorise": (I, [I, I, I, I, I, I, I, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "gcgcn_adam_step": (I, [I, P, L, ctypes.c_double, ctypes.c_double, ctypes.c_double, P]),
    "gcgcn_gemm": (I, [I, I, I, P, L, I, P, L, I, P, L, I, L, L, L, F, P, I, I, I, I, P, L, P]),
    "gcgcn_gemm_dyn": (I, [I, I, I, P, L, I, P, L, I, P, L, P, I, P, I, L, P, L, P]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load the HIP library once.  Raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"gcgcn_amd: {LIB_PATH} not found. Build it with `make -C gcgcn_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.gcgcn_version() != ABI_VERSION:
            raise RuntimeError(f"gcgcn_amd: ABI version {handle.gcgcn_version()} != {ABI_VERSION}")
        _lib = handle
    return _lib


def call(name: str, *args):
    """Invoke an int-status entry point; raise RuntimeError with the library's message on failure."""
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {h.gcgcn_last_error().decode()}")


def layout(kind: str, *dims) -> list:
    """Offsets (in floats) of a block's flat parameter buffer, straight from the library."""
    n = {"gat": 9, "mha": 3, "gcn": 7, "producer": 17, "head": 7}[kind]
    out = (c_int64 * n)()
    call(f"gcgcn_{kind}_layout", *dims, ctypes.cast(out, c_void_p))
    return list(out)
