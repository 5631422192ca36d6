"""ctypes binding of libvx.so (the C-ABI declared in include/vx_abi.h).

There is no CPU fallback: if the library is missing the import of any op fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VX_LIB", os.path.join(_HERE, "libvx.so"))  # VX_LIB: diagnostic builds only

c_void_p, c_int32, c_int64, c_size_t = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)

VX_F32, VX_BF16 = 0, 1
VX_EPI_RELU = 1
VX_CONV_NO_COMPACT = 0x100
VX_CONV_NO_SPLIT_TAIL = 0x200
VX_CONV_F32_MFMA, VX_CONV_F32_PRODUCTS6, VX_CONV_F32_PRODUCTS9 = 0x1000, 0x2000, 0x3000
ARITH_FLAG = {None: 0, 0: VX_CONV_F32_MFMA, 6: VX_CONV_F32_PRODUCTS6, 9: VX_CONV_F32_PRODUCTS9}

# name -> (restype, argtypes); mirrors include/vx_abi.h one to one
SIGNATURES = {
    "vx_version": (c_int32, []),
    "vx_last_error": (ctypes.c_char_p, []),
    "vx_voxelize_workspace_bytes": (c_size_t, [c_int64]),
    "vx_voxelize_hard": (c_int32, [c_void_p, c_int64, c_int32, c_f32p, c_f32p, c_int32, c_int64, c_int32,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_i64p,
                                   c_void_p, c_size_t, c_void_p]),
    "vx_voxelize_hard_batch": (c_int32, [c_void_p, c_int32, c_i64p, c_int32, c_f32p, c_f32p, c_int32, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_i64p,
                                         c_void_p, c_size_t, c_void_p]),
    "vx_voxelize_static_max_points": (c_int64, []),
    "vx_voxelize_static_init": (c_int32, []),
    "vx_voxelize_hard_static": (c_int32, [c_void_p, c_int64, c_int32, c_f32p, c_f32p, c_int32, c_int64, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_i64p,
                                          c_void_p]),
    "vx_rulebook_workspace_bytes": (c_size_t, [c_int64, c_int32, c_i32p]),
    "vx_rulebook_subm": (c_int32, [c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_void_p, c_int64,
                                   c_void_p, c_size_t, c_void_p]),
    "vx_rulebook_conv_count": (c_int32, [c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_i32p, c_i32p,
                                         c_i32p, c_i64p, c_void_p, c_size_t, c_void_p]),
    "vx_rulebook_conv_emit": (c_int32, [c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_i32p, c_i32p,
                                        c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_rank_table_bytes": (c_size_t, [c_int32, c_i32p]),
    "vx_rulebook_subm_ranked": (c_int32, [c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_void_p, c_void_p,
                                          c_int64, c_void_p]),
    "vx_pairs_workspace_bytes": (c_size_t, [c_int32, c_int64]),
    "vx_nbr_to_pairs": (c_int32, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "vx_spconv_fwd": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "vx_spconv_pack_weight_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "vx_spconv_pack_weight": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "vx_rulebook_sort_rows": (c_int32, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_int64, c_void_p,
                                        c_void_p]),
    "vx_spconv_dgrad": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                  c_int64, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "vx_spconv_wgrad_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int64]),
    "vx_spconv_wgrad": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p,
                                  c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "vx_spconv_set_f32_products": (c_int32, [c_int32]),
    "vx_spconv_get_f32_products": (c_int32, []),
    "vx_to_dense": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_i32p, c_void_p,
                              c_int32, c_void_p]),
    "vx_fill": (c_int32, [c_void_p, c_int32, c_size_t, c_void_p]),
    "vx_voxelize_hard_batch_dn": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_f32p, c_f32p, c_int32,
                                            c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                            c_void_p]),
    "vx_rulebook_subm_dn": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_void_p, c_int64,
                                      c_void_p, c_size_t, c_void_p]),
    "vx_rulebook_conv_dn": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "vx_rulebook_subm_ranked_dn": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_i32p, c_i32p, c_void_p,
                                             c_void_p, c_int64, c_void_p]),
    "vx_spconv_fwd_dn": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                   c_int32, c_void_p]),
    "vx_from_dense": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_i32p, c_void_p, c_int32, c_void_p]),
    "vx_to_dense_dn": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_i32p, c_void_p,
                                 c_int32, c_void_p]),
    "vx_bn_workspace_bytes": (c_size_t, [c_int32]),
    "vx_bn_stats": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_bn_finalize": (c_int32, [c_void_p, c_int32, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "vx_bn_apply": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                              c_void_p, c_void_p, c_void_p]),
    "vx_bn_bwd_reduce": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "vx_bn_bwd_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vx_boxes_overlap_bev": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "vx_nms_workspace_bytes": (c_size_t, [c_int32]),
    "vx_nms_bev": (c_int32, [c_void_p, c_int32, ctypes.c_float, c_int32, c_void_p, c_i64p, c_void_p, c_size_t,
                             c_void_p]),
}

_lib = None

# VX_TRACE=1: one line on stderr BEFORE every C-ABI call (entry point, "[capturing]" when the current stream is being
# captured), flushed at once, so that a fatal failure below Python (a HIP runtime abort, a GPU fault) names the call
# it happened in. Off by default: nothing is wrapped and calls go straight to the ctypes functions.
TRACE = os.environ.get("VX_TRACE", "0") not in ("", "0")


def trace(msg):
    if TRACE:
        import sys
        sys.stderr.write(f"[vx] {msg}\n")
        sys.stderr.flush()


class _Traced:
    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name in ("vx_last_error", "vx_version") or name.endswith("_bytes") or "f32_products" in name:
            return fn

        def call(*args):
            import torch
            cap = " [capturing]" if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing() else ""
            trace(f"{name}{cap}")
            return fn(*args)
        return call


class VxError(RuntimeError):
    pass


def lib():
    """Load libvx.so once. Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VxError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vista_amd/csrc`. There is no CPU fallback for this path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = _Traced(L) if TRACE else L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vx_last_error()
        raise VxError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def i32x3(v):
    if isinstance(v, int):
        v = [v, v, v]
    v = [int(x) for x in v]
    if len(v) != 3:
        raise ValueError(f"expected an int or 3 ints, got {v}")
    return (ctypes.c_int32 * 3)(*v)
