"""ctypes binding of libnst_hip.so (C ABI: include/nst_hip.h).

The shared library is the product: there is no CPU or eager-PyTorch fallback.  If it is missing
or does not export every symbol the header declares, importing the hot path fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NST_LIB") or os.path.join(_HERE, "libnst_hip.so")   # NST_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")

NST_OK = 0
NST_VGG19_CONVS = 13
NST_MAX_LEVELS = 8
NST_LOSS_ROW = 4
NST_OPT_ADAM = 0
NST_OPT_LBFGS = 1

c_float_p = C.POINTER(C.c_float)
REDUCE_HOOK = C.CFUNCTYPE(None, C.c_void_p)
c_void = C.c_void_p


NST_CONV_F32, NST_CONV_BF16X3, NST_CONV_F16X2 = 0, 1, 2
CONV_MODES = {"f32": NST_CONV_F32, "bf16x3": NST_CONV_BF16X3, "f16x2": NST_CONV_F16X2}
NST_COMM_ID_BYTES = 128


class StepInfo(C.Structure):
    _fields_ = [("closures", C.c_int), ("total_closures", C.c_int), ("accepted", C.c_int),
                ("loss", C.c_float), ("lr", C.c_float), ("t", C.c_float), ("history", C.c_int)]


class Options(C.Structure):
    """nst_options: -1 = take the environment variable (read once at context creation), else the default."""
    _fields_ = [("struct_size", C.c_int), ("conv_mode", C.c_int), ("batched", C.c_int), ("single_stream", C.c_int),
                ("use_graph", C.c_int), ("h2_band_rows", C.c_int), ("lbfgs_gram", C.c_int), ("h2_mfma16", C.c_int),
                ("h2_wg256", C.c_int), ("h2_tile_rows", C.c_int), ("gram_overlap", C.c_int), ("h2_persist", C.c_int), ("level_split", C.c_int), ("h2_winograd", C.c_int)]


# name -> (restype, argtypes); every symbol include/nst_hip.h declares
SYMBOLS = {
    "nst_version": (C.c_int, []),
    "nst_last_error": (C.c_char_p, [c_void]),
    "nst_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nst_ctx_create": (C.c_int, [C.c_int, C.POINTER(c_void), C.POINTER(c_void), C.POINTER(c_void)]),
    "nst_options_default": (None, [C.POINTER(Options)]),
    "nst_ctx_create_ex": (C.c_int, [C.c_int, C.POINTER(c_void), C.POINTER(c_void), C.POINTER(Options), C.POINTER(c_void)]),
    "nst_ctx_destroy": (None, [c_void]),
    "nst_job_configure": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int]),
    "nst_level_set_targets": (C.c_int, [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, c_void]),
    "nst_closure": (C.c_int, [c_void, c_void, C.c_float, C.c_float, C.c_float, c_void, c_void, c_void]),
    "nst_closure_levels": (C.c_int, [c_void, c_void, C.c_float, C.c_float, C.c_float, C.c_uint, c_void, c_void, c_void]),
    "nst_opt_shard_levels": (C.c_int, [c_void, C.c_uint, c_void, c_void, c_void, c_void]),
    "nst_opt_shard_levels_comm": (C.c_int, [c_void, C.c_uint, c_void]),
    "nst_opt_history": (C.c_int, [c_void, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nst_comm_unique_id": (C.c_int, [c_void]),
    "nst_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, c_void, C.POINTER(c_void)]),
    "nst_comm_destroy": (None, [c_void]),
    "nst_comm_info": (C.c_int, [c_void, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "nst_comm_allreduce_sum": (C.c_int, [c_void, c_void, C.c_size_t, c_void]),
    "nst_adam_step": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_size_t, C.c_int, C.c_double, c_void]),
    "nst_lbfgs_direction": (C.c_int, [c_void, c_void, C.POINTER(c_void), C.POINTER(c_void), C.POINTER(C.c_float), C.c_int,
                                      C.c_float, C.c_size_t, C.c_int, c_void, c_void]),
    "nst_opt_create": (C.c_int, [c_void, C.c_int, C.c_float, C.c_int, C.POINTER(c_void)]),
    "nst_opt_destroy": (None, [c_void]),
    "nst_opt_step": (C.c_int, [c_void, c_void, C.c_float, C.c_float, C.c_float, c_void, C.c_int,
                               C.POINTER(StepInfo), c_void]),
    "nst_vgg_features": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.POINTER(c_void), c_void]),
    "nst_vgg_features_backward": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.POINTER(c_void), c_void, c_void]),
    "nst_vgg_activations": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.POINTER(c_void), c_void]),
    "nst_level_activation": (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void]),
    "nst_level_image": (C.c_int, [c_void, C.c_int, c_void, c_void]),
    "nst_gram": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nst_total_variation": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void]),
    "nst_bicubic_half": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nst_bicubic_half_backward": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nst_prepare_img": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, c_void]),
    "nst_unprepare_img": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, c_void]),
    "nst_resize_bicubic": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, C.c_int, C.c_int, c_void]),
    "nst_gather_rows": (C.c_int, [c_void, c_void, c_void, C.c_size_t, C.c_int, c_void, c_void]),
    "nst_gaussian_mask_accumulate": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                               C.c_double, c_void]),
    "nst_noise_blend": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_double, c_void, c_void]),
    "nst_scale": (C.c_int, [c_void, c_void, C.c_float, C.c_size_t, c_void, c_void]),
    "nst_window_sums_count": (C.c_int, [C.POINTER(C.c_size_t)]),
    "nst_window_begin": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nst_window_end": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, c_void, c_void,
                                 c_void, c_void]),
    "nst_conv_mode": (C.c_int, [c_void]),
    "nst_ctx_bytes": (C.c_int, [c_void, C.POINTER(C.c_size_t)]),
    "nst_set_timing": (C.c_int, [c_void, C.c_int]),
    "nst_last_closure_ms": (C.c_int, [c_void, C.POINTER(C.c_float)]),
    "nst_last_closure_class": (C.c_int, [c_void, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                         C.POINTER(C.c_double)]),
    "nst_dump_last_closure": (C.c_int, [c_void]),
    "nst_timing_totals": (C.c_int, [c_void, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long),
                                    C.POINTER(C.c_double), C.c_int]),
    "nst_timing_mfma_flops": (C.c_int, [c_void, C.c_int, C.POINTER(C.c_double)]),
}

_lib = None
_lock = threading.Lock()


class NstError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into libnst_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True)
    return LIB_PATH


def load():
    """Returns the ctypes handle; raises NstError when the native library is unavailable."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NstError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of the style-transfer "
                f"hot path (no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {CSRC}`.")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NstError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise NstError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(ctx, code: int, what: str) -> None:
    if code != NST_OK:
        msg = load().nst_last_error(ctx)
        raise NstError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
