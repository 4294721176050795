"""ctypes binding of libmcle.so (the C ABI declared in include/mcle.h).

There is no CPU fallback: every compute entry point needs the HIP library and a gfx950
device, and raises :class:`McleError` otherwise.
"""
import ctypes
import importlib.util
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_longlong, c_size_t,
                    c_uint32, c_uint64, c_void_p)

import numpy as np

MCLE_F32, MCLE_F64 = 0, 1
DEMOD_MINDIST, DEMOD_QAM_SLICER = 0, 1
CONST_GENERIC, CONST_QAM, CONST_BPSK = 0, 1, 2
MAX_TAPS = 24
STREAM_DATA, STREAM_NOISE, STREAM_CHAN, STREAM_PHASE = 0, 1, 2, 3
# mcle_ctx_set_option keys (include/mcle.h MCLE_OPT_*): kernel selection for A/B runs and kernel-vs-kernel tests
OPTIONS = {"no_mfma": 0, "mfma_variant": 1, "grid_oversub": 2, "flat_wgs_per_cu": 3, "single_tdl": 4,
           "tdl_mfma_waves": 5, "jakes_direct": 6, "f64_generic": 7,
           "f64_threads": 8, "bd_runtime_solve": 9, "demod_nocert": 10, "f64_variant": 11, "f32_mfma": 12, "tdl_kernel": 13,
           "mimo_tdl_kernel": 14, "walk_legacy": 15}

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# MCLE_LIBRARY: another build of the same library (A/B runs of two builds on one box, scripts/experiments/)
LIB_PATH = os.environ.get("MCLE_LIBRARY") or os.path.join(LIB_DIR, "libmcle.so")


class McleError(RuntimeError):
    """Any failure reported by libmcle (message from mcle_last_error)."""


class McleUnsupported(McleError):
    """A valid request outside a fused kernel's envelope (MCLE_E_UNSUPPORTED): use the staged operators."""


class Counters(Structure):
    _fields_ = [(n, c_uint64) for n in ("n_realizations", "n_skipped", "sym_errors", "sym_errors_sq",
                                        "bit_errors", "bit_errors_sq", "n_symbols", "n_bits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


COUNTER_FIELDS = [n for n, _ in Counters._fields_]


class AwgnCfg(Structure):
    _fields_ = [("n_symbols", c_int32), ("demod_method", c_int32), ("noise_var", c_double)]


class FlatCfg(Structure):
    _fields_ = [("n_symbols", c_int32), ("demod_method", c_int32), ("noise_var", c_double),
                ("Fd", c_double), ("Ts", c_double), ("L", c_int32), ("rayleigh_iid", c_int32)]


class OfdmTdlCfg(Structure):
    _fields_ = [("fft_size", c_int32), ("cp_size", c_int32), ("num_used", c_int32), ("n_ofdm_sym", c_int32),
                ("demod_method", c_int32), ("n_taps", c_int32), ("L", c_int32), ("reserved", c_int32),
                ("noise_var", c_double), ("Fd", c_double), ("Ts", c_double),
                ("tap_power", c_double * MAX_TAPS), ("tap_delay", c_int32 * MAX_TAPS)]


class MimoOfdmCfg(Structure):
    _fields_ = [("nt", c_int32), ("nr", c_int32), ("fft_size", c_int32), ("cp_size", c_int32),
                ("num_used", c_int32), ("n_ofdm_sym", c_int32), ("demod_method", c_int32), ("mmse", c_int32),
                ("noise_var", c_double)]


class MimoOfdmTdlCfg(Structure):
    _fields_ = [("nt", c_int32), ("nr", c_int32), ("fft_size", c_int32), ("cp_size", c_int32),
                ("num_used", c_int32), ("n_ofdm_sym", c_int32), ("demod_method", c_int32), ("mmse", c_int32),
                ("n_taps", c_int32), ("L", c_int32),
                ("noise_var", c_double), ("Fd", c_double), ("Ts", c_double),
                ("tap_power", c_double * MAX_TAPS), ("tap_delay", c_int32 * MAX_TAPS)]


class MimoFlatCfg(Structure):
    _fields_ = [("scheme", c_int32), ("nt", c_int32), ("nr", c_int32), ("n_symbols", c_int32),
                ("demod_method", c_int32), ("mmse", c_int32), ("noise_var", c_double)]


MIMO_SCHEMES = {"blast": 0, "mrc": 1, "mrt": 2, "alamouti": 3, "svd": 4, "gmd": 5}


class IaCfg(Structure):
    _fields_ = [("K", c_int32), ("nr", c_int32), ("nt", c_int32), ("ns", c_int32), ("n_symbols", c_int32),
                ("demod_method", c_int32), ("noise_var", c_double), ("solver", c_int32),
                ("max_iterations", c_int32), ("relative_factor", c_double), ("initialize_with", c_int32),
                ("reserved", c_int32)]


class BdCfg(Structure):
    _fields_ = [("K", c_int32), ("nr", c_int32), ("n_symbols", c_int32), ("demod_method", c_int32),
                ("waterfilling", c_int32), ("has_pathloss", c_int32), ("iPu", c_double), ("noise_var", c_double),
                ("bd_noise_var", c_double), ("pathloss", c_double * 16)]


class BdExtIntCfg(Structure):
    _fields_ = [("num_users", c_int32), ("n_ant_per_user", c_int32), ("n_ext", c_int32), ("method", c_int32),
                ("metric", c_int32), ("num_streams", c_int32), ("ns_user", c_int32 * 4), ("iPu", c_double),
                ("noise_var", c_double), ("pe", c_double)]


BD_METRICS = {None: 0, "None": 0, "naive": 1, "fixed": 2, "capacity": 3, "candidates": 4, "per_user": 5}


class MuStatsCfg(Structure):
    _fields_ = [("K", c_int32), ("n_ext", c_int32), ("joint", c_int32), ("reserved", c_int32), ("nr", c_int32 * 4),
                ("nt", c_int32 * 4), ("ns", c_int32 * 4), ("noise_var", c_double), ("pe", c_double)]


class IaGeneralCfg(Structure):
    _fields_ = [("K", c_int32), ("nr", c_int32), ("nt", c_int32), ("ns", c_int32 * 4), ("solver", c_int32),
                ("initialize_with", c_int32), ("max_iterations", c_int32), ("stream_selection", c_int32),
                ("reserved", c_int32), ("noise_var", c_double), ("relative_factor", c_double)]


IA_STREAM_SELECTION = {None: 0, "none": 0, "greedy": 1, "brute": 2}
IA_INITS = {"random": 0, "fix": 0, "closed_form": 1, "alt_min": 2, "svd": 3}
IA_SOLVERS = {"closed_form": 0, "alt_min": 1, "min_leakage": 2, "max_sinr": 3, "mmse": 4}


class LegacySeg(Structure):
    _fields_ = [("kind", c_int32), ("n", c_int32), ("range", c_uint32), ("reserved", c_uint32)]


_P = c_void_p
_PROTOS = {
    "mcle_last_error": (c_char_p, []),
    "mcle_version": (c_int, []),
    "mcle_device_count": (c_int, [POINTER(c_int)]),
    "mcle_ctx_create": (c_int, [c_int, POINTER(_P)]),
    "mcle_ctx_destroy": (c_int, [_P]),
    "mcle_ctx_set_stream": (c_int, [_P, _P]),
    "mcle_ctx_get_stream": (c_int, [_P, POINTER(_P)]),
    "mcle_ctx_sync": (c_int, [_P]),
    "mcle_ctx_trim_scratch": (c_int, [_P]),
    "mcle_ctx_set_option": (c_int, [_P, c_int, c_longlong]),
    "mcle_ctx_get_option": (c_int, [_P, c_int, POINTER(c_longlong)]),
    "mcle_ctx_device_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "mcle_malloc": (c_int, [_P, c_size_t, POINTER(_P)]),
    "mcle_free": (c_int, [_P, _P]),
    "mcle_memset": (c_int, [_P, _P, c_int, c_size_t]),
    "mcle_memcpy_h2d": (c_int, [_P, _P, _P, c_size_t]),
    "mcle_memcpy_d2h": (c_int, [_P, _P, _P, c_size_t]),
    "mcle_memcpy_2d": (c_int, [_P, _P, c_size_t, _P, c_size_t, c_size_t, c_size_t]),
    "mcle_hbm_stream_rate": (c_int, [_P, c_int, c_size_t, c_int, c_int, POINTER(c_double)]),
    "mcle_timer_start": (c_int, [_P]),
    "mcle_timer_stop_ms": (c_int, [_P, POINTER(c_float)]),
    "mcle_comm_load": (c_int, [c_char_p]),
    "mcle_comm_unique_id": (c_int, [_P, c_size_t]),
    "mcle_comm_init": (c_int, [_P, _P, c_int, c_int]),
    "mcle_comm_destroy": (c_int, [_P]),
    "mcle_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "mcle_counters_allreduce": (c_int, [_P, _P, c_int]),
    "mcle_allreduce_f64": (c_int, [_P, _P, c_size_t]),
    "mcle_set_constellation": (c_int, [_P, POINTER(c_double), c_int, c_int]),
    "mcle_build_demod_grid": (c_int, [POINTER(c_double), c_int, POINTER(c_int), POINTER(c_double), POINTER(c_double),
                                      POINTER(c_double), POINTER(c_uint64)]),
    "mcle_modulate": (c_int, [_P, c_int, _P, _P, c_size_t]),
    "mcle_demodulate": (c_int, [_P, c_int, c_int, _P, _P, c_size_t]),
    "mcle_count_errors": (c_int, [_P, _P, _P, c_size_t, c_size_t, c_int, _P, _P, _P]),
    "mcle_demod_count": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, c_size_t, _P, _P, _P]),
    "mcle_demod_count_u8": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, c_size_t, _P, _P, _P]),
    "mcle_randn_c": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint32, c_uint64, c_double, _P, c_size_t]),
    "mcle_rand_symbols": (c_int, [_P, c_uint64, c_uint64, c_uint64, c_int, _P, c_size_t]),
    "mcle_awgn_add": (c_int, [_P, c_int, _P, _P, c_double, _P, c_size_t]),
    "mcle_jakes_generate": (c_int, [_P, c_int, POINTER(c_double), POINTER(c_double), c_int, c_int, c_double,
                                    c_double, c_double, POINTER(c_double), _P, c_size_t]),
    "mcle_jakes_generate_at": (c_int, [_P, c_int, POINTER(c_double), POINTER(c_double), c_int, c_int, c_double,
                                       POINTER(c_double), POINTER(c_double), _P, c_size_t]),
    "mcle_cmul": (c_int, [_P, c_int, _P, _P, _P, c_size_t]),
    "mcle_tdl_apply": (c_int, [_P, c_int, _P, _P, POINTER(c_int32), c_int, _P, c_size_t]),
    "mcle_tdl_apply_mimo": (c_int, [_P, c_int, _P, _P, POINTER(c_int32), c_int, c_int, c_int, _P, c_size_t,
                                    c_size_t]),
    "mcle_tdl_mean_freq_response": (c_int, [_P, c_int, _P, POINTER(c_int32), c_int, c_int, c_size_t, c_int, c_int,
                                            c_int, _P, c_size_t]),
    "mcle_jakes_taps_philox": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint64, c_int, c_int, c_double, c_double,
                                       c_double, POINTER(c_double), _P, c_size_t]),
    "mcle_awgn_philox": (c_int, [_P, c_int, _P, c_uint64, c_uint64, c_uint64, c_size_t, c_double, _P]),
    "mcle_rand_symbols_batch": (c_int, [_P, c_uint64, c_uint64, c_uint64, c_int, _P, c_size_t]),
    "mcle_rand_modulate_batch": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint64, _P, _P, c_size_t]),
    "mcle_rand_modulate_batch_u8": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint64, _P, _P, c_size_t]),
    "mcle_randn_c_batch": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint64, c_uint32, c_size_t, c_double, _P]),
    "mcle_mimo_channel_philox": (c_int, [_P, c_int, _P, _P, c_uint64, c_uint64, c_double, c_int, c_int, c_size_t, _P,
                                         c_size_t]),
    "mcle_blast_decode_per_subcarrier": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_size_t, _P, c_size_t]),
    "mcle_cdiv": (c_int, [_P, c_int, _P, _P, _P, c_size_t]),
    "mcle_ofdm_modulate": (c_int, [_P, c_int, _P, c_size_t, c_int, c_int, c_int, _P, c_size_t]),
    "mcle_ofdm_demodulate": (c_int, [_P, c_int, _P, c_size_t, c_int, c_int, c_int, _P, c_size_t]),
    "mcle_onetap_equalize": (c_int, [_P, c_int, _P, _P, POINTER(c_int32), c_int, c_size_t, c_int, c_int, c_int,
                                     _P]),
    "mcle_blast_encode": (c_int, [_P, c_int, _P, c_int, c_size_t, _P, c_size_t]),
    "mcle_blast_filter": (c_int, [_P, c_int, _P, c_int, c_int, c_double, _P, _P, c_size_t]),
    "mcle_blast_decode": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_size_t, _P, c_size_t]),
    "mcle_mimo_channel": (c_int, [_P, c_int, _P, _P, _P, c_double, c_int, c_int, c_size_t, _P, c_size_t]),
    "mcle_alamouti_encode": (c_int, [_P, c_int, _P, c_size_t, _P, c_size_t]),
    "mcle_alamouti_decode": (c_int, [_P, c_int, _P, _P, c_int, c_size_t, _P, c_size_t]),
    "mcle_mrt_encode": (c_int, [_P, c_int, _P, _P, c_int, c_size_t, _P, c_size_t]),
    "mcle_mrt_decode": (c_int, [_P, c_int, _P, _P, c_int, c_size_t, _P, c_size_t]),
    "mcle_svd_filters": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_size_t]),
    "mcle_gmd_filters": (c_int, [_P, c_int, _P, c_int, c_double, _P, _P, _P, _P, c_size_t]),
    "mcle_bd_extint": (c_int, [_P, POINTER(BdExtIntCfg), _P, _P, _P, _P, _P, _P, c_size_t]),
    "mcle_ia_solve_general": (c_int, [_P, POINTER(IaGeneralCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t]),
    "mcle_mu_link_stats": (c_int, [_P, POINTER(MuStatsCfg), _P, _P, _P, _P, _P, _P, _P, _P, c_size_t]),
    "mcle_post_processing_sinrs": (c_int, [_P, _P, _P, _P, c_double, c_int, c_int, c_int, _P, c_size_t]),
    "mcle_run_awgn": (c_int, [_P, c_int, POINTER(AwgnCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_run_flat_fading": (c_int, [_P, c_int, POINTER(FlatCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_run_ofdm_tdl": (c_int, [_P, c_int, POINTER(OfdmTdlCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_run_mimo_ofdm": (c_int, [_P, c_int, POINTER(MimoOfdmCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_run_mimo_flat": (c_int, [_P, c_int, POINTER(MimoFlatCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_run_mimo_ofdm_tdl": (c_int, [_P, c_int, POINTER(MimoOfdmTdlCfg), c_uint64, c_uint64, c_uint64, _P, _P,
                                       _P]),
    "mcle_run_ia": (c_int, [_P, c_int, POINTER(IaCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P, _P, _P]),
    "mcle_ia_iterative": (c_int, [_P, c_int, c_int, _P, _P, c_double, c_int, c_double, _P, _P, _P, _P, _P, _P,
                                  c_size_t]),
    "mcle_ia_closed_form": (c_int, [_P, _P, c_double, _P, _P, _P, _P, _P, c_size_t]),
    "mcle_waterfilling": (c_int, [_P, _P, c_int, c_double, c_double, _P, _P, c_size_t]),
    "mcle_block_diagonalize": (c_int, [_P, _P, c_int, c_int, c_double, c_double, c_int, _P, _P, _P, _P, _P, c_size_t]),
    "mcle_pinv": (c_int, [_P, _P, c_int, c_int, c_double, _P, c_size_t]),
    "mcle_run_bd": (c_int, [_P, c_int, POINTER(BdCfg), c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "mcle_legacy_draws": (c_int, [_P, POINTER(LegacySeg), c_int, c_uint32, c_uint64, c_uint64, _P, c_size_t, _P,
                                  c_size_t, _P]),
    "mcle_complex_from_parts": (c_int, [_P, c_int, _P, _P, c_double, _P, c_size_t]),
}

_lib = None


def _preload_hip_runtime():
    """Make libmcle share PyTorch's HIP runtime when PyTorch is installed.

    torch ships its own libamdhip64.so (soname libamdhip64.so.7, same as /opt/rocm's).  Loading
    that copy first lets the dynamic loader satisfy libmcle's NEEDED entry by soname, so torch
    tensors, torch streams and libmcle kernels live in ONE runtime.  MCLE_HIP_RUNTIME=system skips it.
    """
    if os.environ.get("MCLE_HIP_RUNTIME", "torch") != "torch":
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None


def load():
    """Load libmcle.so (once) and attach prototypes.  Raises McleError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McleError("libmcle.so is not built (%s missing): run `make -C pyphysim_amd/csrc` or "
                        "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback"
                        % LIB_PATH)
    _preload_hip_runtime()
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise McleError("cannot load %s: %s" % (LIB_PATH, exc))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def torch_rccl_path():
    """Path of the librccl.so PyTorch bundles (so that libmcle and torch.distributed share one RCCL), or None."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    return cand if os.path.exists(cand) else None


def exported_symbols():
    return sorted(_PROTOS)


def check(rc):
    if rc != 0:
        msg = load().mcle_last_error()
        raise (McleUnsupported if rc == -5 else McleError)((msg or b"unknown error").decode("utf-8", "replace"))


def device_count():
    n = c_int(0)
    check(load().mcle_device_count(byref(n)))
    return n.value


def np_complex(dtype):
    return np.complex64 if dtype == MCLE_F32 else np.complex128


def dtype_code(dtype):
    if dtype in (MCLE_F32, "f32", "float32", np.float32, np.complex64, "complex64"):
        return MCLE_F32
    if dtype in (MCLE_F64, "f64", "float64", np.float64, np.complex128, "complex128", complex, float):
        return MCLE_F64
    raise ValueError("dtype must be 'f32' or 'f64' (got %r)" % (dtype,))
