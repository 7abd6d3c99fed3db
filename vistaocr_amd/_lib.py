"""ctypes binding of libvocr.so (the C-ABI declared in include/vocr.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.  The
product path never computes on the CPU and never imports oracle/."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvocr.so")

P, I, F, Z, U64, U32 = c_void_p, c_int, c_float, c_size_t, c_uint64, c_uint32

# name -> (restype, argtypes); mirrors include/vocr.h one to one (tests/test_abi.py checks both directions)
SIGNATURES = {
    "vocr_last_error": (c_char_p, []),
    "vocr_abi_version": (I, []),
    "vocr_device_count": (I, []),
    "vocr_profile_range_push": (I, [c_char_p]),
    "vocr_profile_range_pop": (I, []),
    "vocr_conv3x3_pack_weights": (I, [P, P, P, I, I, P]),
    "vocr_conv3x3_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_wgrad_workspace_bytes": (Z, [I, I, I, I, I]),
    "vocr_conv3x3_wgrad": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_wino_supported": (I, [I, I]),
    "vocr_conv3x3_wino_pack_floats": (Z, [I, I]),
    "vocr_conv3x3_wino_pack_weights": (I, [P, P, P, I, I, P]),
    "vocr_conv3x3_wino_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_wgrad_wino_workspace_bytes": (Z, [I, I, I, I, I]),
    "vocr_conv3x3_wgrad_wino": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_f16_pack_bytes": (Z, [I, I, I]),
    "vocr_conv3x3_f16_pack_weights": (I, [P, P, P, I, I, P]),
    "vocr_conv3x3_f16_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_wgrad_f16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_c1_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_conv3x3_c1_wgrad_workspace_bytes": (Z, [I, I, I]),
    "vocr_conv3x3_c1_wgrad": (I, [P, P, P, P, P, I, I, I, I, P]),
    "vocr_conv3x3_h16_supported": (I, [I, I]),
    "vocr_conv3x3_h16_plan": (I, [I, I, I, I, I]),
    "vocr_conv3x3_h16_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_f16_padded_row": (I, [I]),
    "vocr_f32_to_f16_layouts": (I, [P, P, P, I, I, I, I, P]),
    "vocr_conv3x3_wgrad_h16_supported": (I, [I, I]),
    "vocr_conv3x3_wgrad_h16_workspace_bytes": (Z, [I, I, I, I, I]),
    "vocr_conv3x3_wgrad_h16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vocr_channel_sum_workspace_bytes": (Z, [I, I, I]),
    "vocr_channel_sum": (I, [P, P, I, I, I, P, P]),
    "vocr_bn_workspace_bytes": (Z, [I, I, I]),
    "vocr_bn_train_stats": (I, [P, I, I, I, F, F, P, P, P, P, P, P, P, P]),
    "vocr_bn_eval_stats": (I, [P, P, I, F, P, P, P]),
    "vocr_bn_relu_apply": (I, [P, P, P, P, P, P, I, I, I, P]),
    "vocr_bn_train_relu_apply": (I, [P, P, P, P, I, I, I, F, F, P, P, P, P, P, P, P, P]),
    "vocr_bn_train_relu_fracpool2x2_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, F, P, P, P, P, P, P, P, P]),
    "vocr_bn_relu_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "vocr_bn_relu_fracpool2x2_bwd_supported": (I, [I, I, I, I]),
    "vocr_bn_relu_fracpool2x2_bwd": (I, [P] * 13 + [I] * 6 + [P, P]),
    "vocr_fracpool2x2_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "vocr_bn_relu_fracpool2x2_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "vocr_fracpool2x2_bwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "vocr_relu_maxpool2_fwd": (I, [P, P, P, I, I, I, I, P]),
    "vocr_relu_maxpool2_bwd": (I, [P, P, P, P, I, I, I, I, P]),
    "vocr_gemm_workspace_bytes": (Z, [I, I, I, I]),
    "vocr_gemm": (I, [I, I, I, I, I, P, I, P, I, P, I, P, I, I, P, Z, P]),
    "vocr_gemm_pair_workspace_bytes": (Z, [I, I, I, I]),
    "vocr_gemm_pair": (I, [I, I, I, I, I, I, P, P, I, P, P, I, P, P, I, P, P, I, P, Z, P]),
    "vocr_gemm_x6_planes_bytes": (Z, [I, I]),
    "vocr_gemm_x6_split": (I, [P, P, I, I, P, ctypes.c_long, I, I, I, P, P]),
    "vocr_gemm_x6_workspace_bytes": (Z, [I, I, I]),
    "vocr_gemm_x6": (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, P, P, I, I, I, P, P, I, P, P]),
    "vocr_gemm_x6_two_views": (I, [P, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, P]),
    "vocr_gemm_h3_two_views": (I, [P, I, I, I, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, P]),
    "vocr_gemm_h3_planes_bytes": (Z, [I, I]),
    "vocr_gemm_h3_split": (I, [P, P, I, I, P, ctypes.c_long, I, I, I, ctypes.c_float, P, P]),
    "vocr_gemm_h3": (I, [P, I, I, I, I, P, I, I, I, I, I, I, I, P, P, I, I, I, P, P, I, P, P]),
    "vocr_colsum_workspace_bytes": (Z, [I, I]),
    "vocr_colsum": (I, [P, P, I, I, P, P]),
    "vocr_relu_bwd": (I, [P, P, P, Z, P]),
    "vocr_bchw_to_wbch": (I, [P, P, I, I, I, I, P]),
    "vocr_wbch_to_bchw": (I, [P, P, I, I, I, I, P]),
    "vocr_mul": (I, [P, P, P, Z, P]),
    "vocr_add": (I, [P, P, P, Z, P]),
    "vocr_scale_dev": (I, [P, P, P, Z, P]),
    "vocr_dropout_fwd": (I, [P, P, P, Z, F, U64, P]),
    "vocr_dropout_mask": (I, [P, Z, F, U64, P]),
    "vocr_lstm_workspace_bytes": (Z, [I, I, I]),
    "vocr_lstm_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "vocr_lstm_fwd_range": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "vocr_lstm_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "vocr_lstm_bwd_bias": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "vocr_lstm_bwd_parts_supported": (I, [I, I, I]),
    "vocr_lstm_bwd_parts": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "vocr_lstm_bias_from_parts": (I, [P, P, I, I, I, P]),
    "vocr_lstm_packed_supported": (I, [I, I]),
    "vocr_seq_rowmap": (I, [P, I, I, I, P, P, P]),
    "vocr_gather_rows": (I, [P, P, P, ctypes.c_long, I, P, P]),
    "vocr_gather_rows_fill_grad_workspace_bytes": (Z, [I]),
    "vocr_gather_rows_fill_grad": (I, [P, P, ctypes.c_long, I, P, P, P]),
    "vocr_lstm_fwd_packed": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    "vocr_lstm_follow_supported": (I, [I, I]),
    "vocr_lstm_xproj_pack_bytes": (Z, [I]),
    "vocr_lstm_xproj_pack": (I, [P, P, P, I, P]),
    "vocr_lstm_fwd_lead": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P]),
    "vocr_lstm_bwd_packed": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    "vocr_ctc_workspace_bytes": (Z, [I, I, I, I]),
    "vocr_ctc_loss_grad": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vocr_argmax_rows": (I, [P, P, P, I, I, P]),
    "vocr_greedy_collapse": (I, [P, P, P, P, P, P, I, I, F, P]),
    "vocr_clamp_adam": (I, [P, P, P, P, Z, F, F, F, F, F, F, F, I, P, P]),
    "vocr_clamp": (I, [P, Z, F, P, P]),
    "vocr_comm_unique_id": (I, [P]),
    "vocr_comm_create": (I, [ctypes.POINTER(P), P, I, I, I]),
    "vocr_allreduce_sum_f32": (I, [P, P, Z, P]),
    "vocr_comm_destroy": (I, [P]),
}

ABI_VERSION = 5          # include/vocr.h: VOCR_ABI_VERSION

_lib = None


def load():
    """Load libvocr.so (building is __graft_entry__.build()'s / vistaocr_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libvocr.so not found at %s — run `python -m vistaocr_amd.build` (needs hipcc). "
                           "vistaocr_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.vocr_abi_version.restype = ctypes.c_int
    got = lib.vocr_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("libvocr.so at %s answers ABI version %d, this package was written for %d (include/vocr.h: VOCR_ABI_VERSION) "
                           "— a stale binary; rebuild with `python -m vistaocr_amd.build --force`" % (LIB_PATH, got, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vocr_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


# optional per-entry-point HIP-event timing (bench.py's roofline leg): name -> list of (args, start, end)
_TIMED = None
_FILTER = {}


def enable_timing(names, keep=False):
    """Record a HIP event pair on the current torch stream around every call of the named entry points.  `names`: a list, or
    a dict name -> predicate(args) selecting which calls to time; None switches it off.  keep=True keeps the records gathered
    so far (used to time only some steps of a loop)."""
    global _TIMED, _FILTER
    old = _TIMED if keep and _TIMED is not None else {}
    if not names:
        _TIMED = {k: v for k, v in old.items()} if (keep and old) else None
        _FILTER = {k: (lambda a: False) for k in (_TIMED or {})}
        return
    _FILTER = dict(names) if isinstance(names, dict) else {n: None for n in names}
    _TIMED = {n: old.get(n, []) for n in _FILTER}


def timing_records():
    return _TIMED or {}


def call(name, *args):
    if _TIMED is not None and name in _TIMED and (_FILTER.get(name) is None or _FILTER[name](args)):
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        _TIMED[name].append((args, e0, e1))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)


# ---- named ranges for rocprofv3 --marker-trace (VOCR_ROCTX=1): `with prof_range("lstm.l1"): ...`
import contextlib as _contextlib

ROCTX = os.environ.get("VOCR_ROCTX", "0") == "1"


@_contextlib.contextmanager
def prof_range(name):
    if not ROCTX:
        yield
        return
    lib = load()
    lib.vocr_profile_range_push(name.encode())
    try:
        yield
    finally:
        lib.vocr_profile_range_pop()
