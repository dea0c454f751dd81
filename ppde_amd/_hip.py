"""ctypes binding of the C ABI in include/ppde_hip.h (the only way the package reaches the GPU).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

import torch  # imported first on purpose: the library then binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPDE_HIP_LIB") or os.path.join(_HERE, "libppde_hip.so")   # PPDE_HIP_LIB: tuning builds only

OK, ERR_INVALID, ERR_HIP, ERR_NOT_ONEHOT, ERR_NUMERIC = 0, -1, -2, -3, -4


class PpdeHipError(RuntimeError):
    pass


class ChainConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "n_chains", "max_steps", "pas_length", "nmut_threshold", "paper_results", "min_pos", "max_pos", "which",
        "rng_mode", "reuse_grad", "record_after_reset", "trace", "random_chain", "use_graph", "n_streams")] + [("_pad", C.c_int32)] + \
        [("seed", C.c_uint64), ("chain_offset", C.c_uint64)]


class TfWeights(C.Structure):
    """ppde_tf_weights (include/ppde_hip.h): host pointers to ESM-2's fp32 parameters."""
    _PER_LAYER = ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
                  "fc1_w", "fc1_b", "fc2_w", "fc2_b")
    _GLOBAL = ("final_ln_w", "final_ln_b", "head_dense_w", "head_dense_b", "head_ln_w", "head_ln_b", "head_bias")
    _fields_ = [("embed", C.c_void_p)] + [(k, C.POINTER(C.c_void_p)) for k in _PER_LAYER] + [(k, C.c_void_p) for k in _GLOBAL]


_p, _i, _f = C.c_void_p, C.c_int, C.c_float
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/ppde_hip.h one to one
SIGNATURES = {
    "ppde_abi_version": (_i, []),
    "ppde_last_error": (C.c_char_p, []),
    "ppde_device_count": (_i, []),
    "ppde_model_create": (_i, [_pp, _i, _i, _p]),
    "ppde_model_destroy": (_i, [_p]),
    "ppde_model_set_potts": (_i, [_p, _p, _p, _i, _i]),
    "ppde_model_set_cnn": (_i, [_p, _i, _i, _i, _i, _pp, _pp, _pp, _pp, _pp, _pp]),
    "ppde_model_set_lamda": (_i, [_p, _f]),
    "ppde_model_set_transformer": (_i, [_p, _i, _i, _i, _i, C.POINTER(TfWeights)]),
    "ppde_model_get_transformer_wt_score": (_i, [_p, C.POINTER(_f)]),
    "ppde_debug_transformer_read": (_i, [_p, _i, _i, _p, C.c_int64]),
    "ppde_transformer_time_gemm": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_f)]),
    "ppde_transformer_time_fc1_in_situ": (_i, [_p, _p, _i, C.POINTER(_f), C.POINTER(_i)]),
    "ppde_model_get_wt_hamiltonian": (_i, [_p, C.POINTER(_f)]),
    "ppde_onehot_to_idx": (_i, [_p, _p, _i, _p, _p]),
    "ppde_idx_to_onehot": (_i, [_p, _p, _i, _p, _p]),
    "ppde_energy_grad": (_i, [_p, _p, _i, _i, _p, _p, _p, _p]),
    "ppde_chains_create": (_i, [_pp, _p, C.POINTER(ChainConfig)]),
    "ppde_chains_destroy": (_i, [_p]),
    "ppde_chains_init": (_i, [_p, _p]),
    "ppde_chains_run": (_i, [_p, _i, _p, _p, _p, _p]),
    "ppde_chains_sync": (_i, [_p]),
    "ppde_chains_steps_done": (_i, [_p]),
    "ppde_chains_peek": (_i, [_p, _p, _p, _p, _p, _p]),
    "ppde_chains_collect": (_i, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "ppde_chains_trace": (_i, [_p, _p, _p, _p, _p]),
    "ppde_chains_graph_stats": (_i, [_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ppde_chains_philox_dump": (_i, [_p, _i, _i, _p, _p, _p]),
    "ppde_chains_time_potts_kernel": (_i, [_p, _i, C.POINTER(_f)]),
    "ppde_chains_time_experts": (_i, [_p, _i, C.POINTER(_f)]),
    "ppde_chains_time_potts_in_situ": (_i, [_p, _i, C.POINTER(_f), C.POINTER(_i), C.POINTER(_f)]),
}

_lib = None


def load():
    """Load libppde_hip.so (built by ppde_amd/build.py). Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PpdeHipError(
            f"{LIB_PATH} is missing: build the HIP library first (python -m ppde_amd.build); "
            "ppde_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.ppde_abi_version() != 1:
        raise PpdeHipError("libppde_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().ppde_last_error().decode()
        if rc == ERR_NOT_ONEHOT or rc == ERR_NUMERIC:
            raise ValueError(msg)        # what torch.distributions' validation raises in the reference
        raise PpdeHipError(f"[{rc}] {msg}")


def ptr(t):
    """Raw pointer of a torch tensor / numpy array (or None)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
