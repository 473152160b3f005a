"""ctypes binding of libfact_hip.so: the drop-in C ABI declared in include/fact_hip.h plus the test / bench surface of
include/fact_hip_debug.h (DEBUG_SYMBOLS below: single-op entry points, probes, the kernel-class recorder, A/B knobs).

There is deliberately no CPU fallback: if the HIP library is missing or does not load, importing
this module's `lib()` raises.  The oracle under /oracle is test infrastructure and is never
imported from the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same objects (mint_amd/csrc/build.sh):
#   libfact_hip.so      the production library - exports include/fact_hip.h and nothing else (-fvisibility=hidden)
#   libfact_hip_dbg.so  engine.hip / probe.hip compiled -DFACT_DEBUG_ABI: additionally exports include/fact_hip_debug.h
# A process binds ONE of them (kernel-selection knobs are process-wide state of the library): FACT_DEBUG_ABI=1 in the
# environment before this module is imported selects the test / bench build (tests/conftest.py and bench.py set it).
# FACT_LIB: a development build of the library (tools/build_variant.sh, same-box A/B of kernel variants).  A missing file is
# an error, never a fallback.
DEBUG_ABI = os.environ.get("FACT_DEBUG_ABI", "0") not in ("", "0")
PROD_LIB_PATH = os.path.join(_HERE, "lib", "libfact_hip.so")
DEBUG_LIB_PATH = os.path.join(_HERE, "lib", "libfact_hip_dbg.so")
LIB_PATH = os.environ.get("FACT_LIB") or (DEBUG_LIB_PATH if DEBUG_ABI else PROD_LIB_PATH)

# epilogue kinds (mint_amd/csrc/gemm.h)
EPI_BF16, EPI_F32_BIAS, EPI_F32_BIAS_POS, EPI_F32_BIAS_RESID = 0, 1, 2, 3
EPI_BIAS_GELU, EPI_GELU_BWD, EPI_HEADS, EPI_ATOMIC_F32, EPI_F32_BF16 = 4, 5, 6, 7, 8


class FactStackCfg(C.Structure):
    _fields_ = [("seq_len", C.c_int), ("feature_dim", C.c_int), ("hidden", C.c_int),
                ("layers", C.c_int), ("heads", C.c_int), ("ff", C.c_int)]


class FactConfig(C.Structure):
    _fields_ = [("motion", FactStackCfg), ("audio", FactStackCfg), ("cross", FactStackCfg),
                ("out_dim", C.c_int), ("ln_eps", C.c_float)]


class FactParamDesc(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", C.c_size_t), ("rows", C.c_int),
                ("cols", C.c_int), ("kind", C.c_int)]


class FactArenas(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p),
                ("adam_v", C.c_void_p)]


# fact_grad_cb: void (*)(void* user, int bucket, size_t offset_floats, size_t count_floats)
GRAD_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t)

# name -> (restype, argtypes); kept in sync with include/fact_hip.h + fact_hip_debug.h (tests check every symbol)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "fact_abi_version": (_i, []),
    "fact_crc32c": (C.c_uint, [_vp, C.c_size_t, C.c_uint]),
    "fact_last_error": (C.c_char_p, []),
    "fact_arena_size": (_i, [C.POINTER(FactConfig), C.POINTER(_sz), C.POINTER(_i)]),
    "fact_create": (_i, [C.POINTER(FactConfig), _i, _i, C.POINTER(FactArenas), C.POINTER(_vp)]),
    "fact_destroy": (_i, [_vp]),
    "fact_param_table": (_i, [_vp, C.POINTER(C.POINTER(FactParamDesc)), C.POINTER(_i)]),
    "fact_arenas": (_i, [_vp, C.POINTER(FactArenas), C.POINTER(_sz)]),
    "fact_refresh_weights": (_i, [_vp, _vp]),
    "fact_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "fact_forward_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "fact_adam_step": (_i, [_vp, _f, _f, _f, _f, _f, _vp]),
    "fact_clip_gradients": (_i, [_vp, _f, _vp, _vp]),
    "fact_adam_begin": (_i, [_vp, _f, _f, _f, _f]),
    "fact_adam_bucket": (_i, [_vp, _i, _vp]),
    "fact_adam_bucket_bf16": (_i, [_vp, _i, _vp, _vp]),
    "fact_adam_cancel": (_i, [_vp]),
    "fact_num_buckets": (_i, [_vp, C.POINTER(_i)]),
    "fact_loss": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fact_kprof": (_i, [_vp, _i]),
    "fact_kprof_dump": (_i, [_vp, C.c_char_p]),
    "fact_kprof_read": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fact_kprof_kernels": (_i, [_vp, _i, C.c_char_p, _i]),
    "fact_debug_set_option": (_i, [_vp, C.c_char_p, _i]),
    "fact_get_step": (_i, [_vp, C.POINTER(C.c_int64)]),
    "fact_set_step": (_i, [_vp, C.c_int64]),
    "fact_infer_ar": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, C.POINTER(_i), _vp]),
    "fact_set_option": (_i, [_vp, C.c_char_p, _i]),
    "fact_set_grad_callback": (_i, [_vp, GRAD_CB, _vp, _vp]),
    "fact_cast_f32_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "fact_cast_bf16_f32": (_i, [_vp, _vp, _sz, _vp]),
    "fact_op_gemm_nt": (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i,
                             _vp, _i, _vp]),
    "fact_op_gemm_tn": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "fact_op_gemm_tn_group": (_i, [_i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp),
                                   C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, _vp]),
    "fact_op_gemm_tn_group_cs": (_i, [_i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp),
                                      C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), _i, _vp]),
    "fact_op_gemm_tn_group_adam": (_i, [_i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp),
                                        C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp),
                                        C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, _f, _f, _f, _f, _vp]),
    "fact_op_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "fact_op_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fact_op_attention_scratch": (_sz, [_i, _i, _i, _i]),
    "fact_op_attention": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "fact_op_adam": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    "fact_op_mse": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fact_probe_mfma": (_i, [_vp, _vp, _vp, _vp]),
    "fact_probe_tr": (_i, [_vp, _i, _vp, _vp, _vp]),
    "fact_debug_force_generic_gemm": (_i, [_i]),
    "fact_debug_gemm_nt_variant": (_i, [_i]),
    "fact_debug_gemm_big_impl": (_i, [_i]),
    "fact_debug_gemm_tn_cfg": (_i, [_i]),
    "fact_debug_gemm_splitk_max": (_i, [_i]),
    "fact_debug_gemm_nt_band": (_i, [_i]),
    "fact_debug_ln_bwd": (_i, [_i, _i]),
    "fact_debug_attn_force_tiled": (_i, [_i]),
    "fact_debug_attn_variant": (_i, [_i]),
    "fact_debug_attn_variant_get": (_i, []),
    "fact_debug_cu_hog": (_i, [_i, _i, _vp]),
}

# declared in include/fact_hip_debug.h (everything else: include/fact_hip.h, the drop-in boundary)
DEBUG_SYMBOLS = frozenset(n for n in SIGNATURES if n.startswith(("fact_debug_", "fact_op_", "fact_probe_", "fact_kprof")))
# keys fact_set_option accepts; every other engine knob goes through fact_debug_set_option
PUBLIC_OPTIONS = ("sr_rows", "grad_overwrite", "adam_in_wgrad", "side_stream", "aux_stream")

_LIB = None


class FactError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfact_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load (once) and return the bound library. Raises if the HIP extension is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                " (there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm first: it brings its own HIP runtime, and the library must bind to THAT instance (loaded the
        # other way round the process ends up with two runtimes and the engine sees "no ROCm-capable device")
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if name in DEBUG_SYMBOLS and not DEBUG_ABI:
                    continue  # the production library does not export the test / bench surface
                raise
            fn.restype = res
            fn.argtypes = args
        _LIB = l
    return _LIB


def need_debug_abi(what):
    """Raise a readable error when a test / bench entry point is asked of the production library."""
    if not hasattr(lib(), "fact_debug_set_option"):
        raise RuntimeError("%s needs the test / bench build of the library: set FACT_DEBUG_ABI=1 before importing mint_amd "
                           "(libfact_hip_dbg.so; the production libfact_hip.so exports include/fact_hip.h only)" % what)


def check(rc):
    if rc != 0:
        msg = lib().fact_last_error()
        raise FactError(rc, msg.decode() if msg else "")
    return rc


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
