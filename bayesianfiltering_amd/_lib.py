"""ctypes binding of the C-ABI (include/bayesfilt.h).

The shared library is built in-tree by ``bayesianfiltering_amd/csrc/Makefile`` (or
``__graft_entry__.build()``).  There is NO fallback: if the library is missing, or a filter is
called without a gfx950 device, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BAYESFILT_HIP_LIB: load another build of the same C-ABI (e.g. one with debug timers compiled in)
LIB_PATH = os.environ.get("BAYESFILT_HIP_LIB") or os.path.join(_HERE, "libbayesfilt_hip.so")

BF_OK, BF_EINVAL, BF_EUNSUPPORTED, BF_EHIP, BF_ENOGPU = 0, -1, -2, -3, -4
HEADER_VERSION = 210  # the BF_VERSION of include/bayesfilt.h these bindings were written against


class BayesFiltError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bayesfilt error {code}: {msg}")
        self.code = code


class bf_stream(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sB", C.c_int64), ("sK", C.c_int64), ("sT", C.c_int64), ("sE", C.c_int64)]


class bf_cstream(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sB", C.c_int64), ("sK", C.c_int64), ("sT", C.c_int64), ("sE", C.c_int64)]


class bf_out_desc(C.Structure):
    _fields_ = [("weights", bf_stream), ("means", bf_stream), ("covs", bf_stream), ("pred_means", bf_stream),
                ("pred_covs", bf_stream), ("loglik", bf_stream), ("coll_mean", bf_stream), ("coll_cov", bf_stream)]


class bf_carry(C.Structure):
    _fields_ = [("w_in", C.c_void_p), ("m_in", C.c_void_p), ("P_in", C.c_void_p),
                ("w_out", C.c_void_p), ("m_out", C.c_void_p), ("P_out", C.c_void_p)]


_FP = C.POINTER(C.c_float)


class bf_lgssm(C.Structure):
    _fields_ = [("n", C.c_int32), ("dq", C.c_int32), ("m", C.c_int32), ("dr", C.c_int32),
                ("A", _FP), ("G", _FP), ("H", _FP), ("D", _FP), ("q0", _FP), ("r0", _FP), ("Q", _FP), ("R", _FP),
                ("Q_steps", C.c_int32), ("R_steps", C.c_int32)]


class bf_model(C.Structure):
    _fields_ = [("dyn_id", C.c_int32), ("emi_id", C.c_int32), ("n", C.c_int32), ("dq", C.c_int32), ("m", C.c_int32),
                ("dr", C.c_int32), ("dyn_theta", _FP), ("n_dyn_theta", C.c_int32), ("emi_theta", _FP),
                ("n_emi_theta", C.c_int32), ("q0", _FP), ("r0", _FP), ("Q", _FP), ("R", _FP), ("flags", C.c_int32),
                ("Q_steps", C.c_int32), ("R_steps", C.c_int32), ("user", C.c_void_p)]


BF_MODEL_PREDICT_FIRST, BF_MODEL_NO_JITTER, BF_MODEL_LEGACY_GSF_COV = 1, 2, 4
BF_FN_USER = 100


class bf_ukf_params(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("kappa", C.c_float)]


class bf_bpf_model(C.Structure):
    _fields_ = [("ssm", bf_model), ("m0", _FP), ("P0", _FP), ("lp_cov", _FP), ("r_eval", _FP), ("lp_theta", _FP),
                ("n_lp_theta", C.c_int32)]


class bf_bpf_carry(C.Structure):
    _fields_ = [("x_in", C.c_void_p), ("w_in", C.c_void_p), ("key_in", C.c_void_p),
                ("x_out", C.c_void_p), ("w_out", C.c_void_p), ("key_out", C.c_void_p)]


class bf_bpf_out(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("w_sB", C.c_int64), ("w_sN", C.c_int64), ("w_sT", C.c_int64),
                ("particles", C.c_void_p), ("x_sB", C.c_int64), ("x_sN", C.c_int64), ("x_sT", C.c_int64),
                ("ancestors", C.c_void_p), ("mean", C.c_void_p), ("ess", C.c_void_p), ("logz", C.c_void_p),
                ("resampled", C.c_void_p)]


# every symbol include/bayesfilt.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "bf_version": (C.c_int, []),
    "bf_abi_check": (C.c_int, [C.c_int32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "bf_last_error": (C.c_char_p, []),
    "bf_device_count": (C.c_int, []),
    "bf_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "bf_set_call_option": (C.c_int, [C.c_char_p, C.c_int]),
    "bf_bytes_per_step": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(bf_out_desc)]),
    "bf_gsf_ekf_f32": (C.c_int, [C.POINTER(bf_model), C.POINTER(bf_cstream), C.POINTER(bf_cstream), C.c_int64, C.c_int64,
                                 C.c_int32, C.POINTER(bf_carry), C.POINTER(bf_out_desc), C.c_void_p]),
    "bf_ugsf_ukf_f32": (C.c_int, [C.POINTER(bf_model), C.POINTER(bf_ukf_params), C.POINTER(bf_cstream), C.POINTER(bf_cstream),
                                  C.c_int64, C.c_int64, C.c_int32, C.POINTER(bf_carry), C.POINTER(bf_out_desc), C.c_void_p]),
    "bf_agsf_ekf_f32": (C.c_int, [C.POINTER(bf_model), C.POINTER(bf_cstream), C.POINTER(bf_cstream), C.c_int64, C.c_int64,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_uint32), _FP, C.POINTER(bf_carry),
                                  C.POINTER(bf_out_desc), C.c_void_p, C.c_int32, C.c_void_p]),
    "bf_agsf_ukf_f32": (C.c_int, [C.POINTER(bf_model), C.POINTER(bf_ukf_params), C.POINTER(bf_cstream), C.POINTER(bf_cstream),
                                  C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), _FP, C.POINTER(bf_carry),
                                  C.POINTER(bf_out_desc), C.c_void_p, C.c_int32, C.c_void_p]),
    "bf_optimal_resample_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "bf_collapse_f32": (C.c_int, [C.POINTER(bf_stream), C.POINTER(bf_stream), C.POINTER(bf_stream), C.c_int64, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_bpf_f32": (C.c_int, [C.POINTER(bf_bpf_model), C.POINTER(bf_cstream), C.POINTER(bf_cstream), C.c_int64, C.c_int64,
                             C.c_int32, C.POINTER(C.c_uint32), C.c_float, C.c_int32, C.POINTER(bf_bpf_carry),
                             C.POINTER(bf_bpf_out), C.c_void_p]),
    "bf_sample_ssm_f32": (C.c_int, [C.POINTER(bf_bpf_model), C.c_void_p, C.POINTER(bf_cstream), C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "bf_resample_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "bf_user_model_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bf_user_model_create_lp": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bf_user_model_destroy": (None, [C.c_void_p]),
    "bf_allgather_summaries": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bf_canon_eval_f32": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "bf_random_normal_f32": (C.c_int, [C.POINTER(C.c_uint32), C.c_int64, _FP]),
    "bf_random_split": (C.c_int, [C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_uint32)]),
    "bf_kalman_filter_f32": (C.c_int, [C.POINTER(bf_lgssm), C.POINTER(bf_cstream), C.c_int64, C.c_int64,
                                       C.POINTER(bf_carry), C.POINTER(bf_out_desc), C.c_void_p]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C bayesianfiltering_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's): it must be the one
    # already loaded when this library binds libamdhip64, or the process ends up with a mixed
    # runtime and torch reports "No HIP GPUs are available".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # the structs above mirror include/bayesfilt.h by hand: let the library compare layouts before anything is launched
    if lib.bf_abi_check(HEADER_VERSION, C.sizeof(bf_out_desc), C.sizeof(bf_lgssm), C.sizeof(bf_model), C.sizeof(bf_bpf_model),
                        C.sizeof(bf_bpf_out)) != BF_OK:
        raise ImportError("ABI mismatch between _lib.py and the built library: " + lib.bf_last_error().decode())
    _lib = lib
    return lib


def check(code):
    if code != BF_OK:
        raise BayesFiltError(code, load().bf_last_error().decode())


def require_gpu():
    lib = load()
    if lib.bf_device_count() < 1:
        raise BayesFiltError(BF_ENOGPU, "no gfx950 (MI355X) device visible; the filters only run on the HIP path")
    return lib


def arm_call_options(lib, options):
    """Per-call overrides of the tuning options (bf_set_call_option): ``options`` is a dict name -> int; they apply to the
    next filter entry point called on this thread and to that call only."""
    if options:
        for k, v in options.items():
            check(lib.bf_set_call_option(k.encode() if isinstance(k, str) else k, int(v)))
