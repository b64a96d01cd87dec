"""ctypes binding of the C ABI in include/pevit_hip.h.

The HIP library is the product: there is no CPU or PyTorch fallback behind these calls.  If
``libpevit_hip.so`` is missing or fails to load, importing the engine raises -- build it with
``python -c 'import __graft_entry__ as g; g.build()'`` (or ``make -C pevit_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpevit_hip.so")

c_void_p, c_int, c_float, c_size_t, c_char_p = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_char_p


class PevitDims(C.Structure):
    _fields_ = [("width", C.c_int32), ("layers", C.c_int32), ("patch", C.c_int32), ("resolution", C.c_int32),
                ("out_dim", C.c_int32), ("method", C.c_int32), ("lora_rank", C.c_int32),
                ("num_classes", C.c_int32), ("weight_format", C.c_int32)]


WEIGHT_FORMATS = {"bf16": 0, "fp8": 1, "f32-verify": 2, "fp8-act": 3}


METHOD_IDS = {"kadaptation": 0, "lora": 1, "adapter": 2, "compacter": 3, "none": 4}

P = c_void_p
# name -> (restype, argtypes); mirrors include/pevit_hip.h one to one
SIGNATURES = {
    "pevit_last_error": (c_char_p, []),
    "pevit_version": (c_int, []),
    "pevit_ctx_create": (c_int, [C.POINTER(PevitDims), C.POINTER(c_void_p)]),
    "pevit_ctx_destroy": (None, [P]),
    "pevit_arena_bytes": (c_size_t, [P]),
    "pevit_workspace_bytes": (c_size_t, [P, c_int]),
    "pevit_num_tower_params": (c_size_t, [P]),
    "pevit_num_params": (c_size_t, [P]),
    "pevit_param_grad_mask": (c_int, [P, P, c_size_t]),
    "pevit_bind": (c_int, [P, P, c_size_t, P, c_size_t, c_int]),
    "pevit_set_params": (c_int, [P, P, P, P, P]),
    "pevit_load_block": (c_int, [P, P, c_int] + [P] * 12),
    "pevit_load_stem": (c_int, [P, P] + [P] * 8),
    "pevit_load_phm_rule": (c_int, [P, P, P]),
    "pevit_transformer_forward": (c_int, [P, P, P, P, c_int, c_int]),
    "pevit_transformer_backward": (c_int, [P, P, P, P, c_int]),
    "pevit_blocks_forward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int]),
    "pevit_blocks_backward": (c_int, [P, P, P, P, c_int, c_int, c_int]),
    "pevit_visual_forward": (c_int, [P, P, P, P, c_int, c_int]),
    "pevit_visual_backward": (c_int, [P, P, P, c_int]),
    "pevit_visual_backward_part": (c_int, [P, P, P, c_int, c_int, c_int]),
    "pevit_param_layer_offset": (c_size_t, [P, c_int]),
    "pevit_head_forward_backward": (c_int, [P, P, P, P, P, P, c_int, P, P, P, c_int]),
    "pevit_zero_grads": (c_int, [P, P]),
    "pevit_sgd_step": (c_int, [P, P, c_float, c_float, c_float, c_float, c_int]),
    "pevit_train_forward_backward": (c_int, [P, P, P, P, P, P, c_int, P, P, c_int]),
    "pevit_set_input_norm": (c_int, [P, P, P]),
    "pevit_visual_forward_u8": (c_int, [P, P, P, P, c_int, c_int]),
    "pevit_train_forward_backward_u8": (c_int, [P, P, P, P, P, P, c_int, P, P, c_int]),
    "pevit_profile_begin": (c_int, [P, c_int]),
    "pevit_profile_end": (c_int, [P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int)]),
    "pevit_profile_launch": (c_int, [P, c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int)]),
    "pevit_profile_launch_bytes": (c_int, [P, c_int, C.POINTER(C.c_double)]),
    "pevit_op_gemm": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int,
                              P, c_int, P, c_int, P, c_int, c_size_t, c_int, c_int, c_int]),
    "pevit_op_gemm_fp8": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, P, c_int, c_int, c_int, P, P, c_int, P, c_int,
                                  P, c_int, P, c_int, P, c_int, c_size_t, c_int, c_int, c_int]),
    "pevit_op_gemm_f8a": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, c_int, c_int, c_int, P, P, c_int, P, c_int, P, c_int,
                                  P, c_int, c_int, c_size_t, c_int, c_int, c_int]),
    "pevit_op_cast_fp8": (c_int, [P, P, P, c_int, c_int]),
    "pevit_debug_last_gemm_path": (c_int, []),
    "pevit_op_quant_fp8": (c_int, [P, P, c_int, c_int, P, P, P]),
    "pevit_op_dequant_fp8": (c_int, [P, P, P, c_int, c_int, P]),
    "pevit_op_ln_fwd": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P]),
    "pevit_op_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int]),
    "pevit_op_ln_bwd_scaled": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "pevit_op_attn_fwd": (c_int, [P, P, P, P, P, c_int, P, c_int, c_int, c_int]),
    "pevit_op_attn_bwd": (c_int, [P, P, P, P, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int]),
    "pevit_op_cast_bf16": (c_int, [P, P, P, c_size_t, c_float]),
    "pevit_op_delta_add": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, c_int]),
    "pevit_op_attn_fwd_delta": (c_int, [P, P, P, P, P, P, P, c_float, P, c_int, P, c_int, c_int, c_int]),
    "pevit_op_attn_delta_hpw": (c_int, [c_int, c_int, c_int]),
    "pevit_debug_timeline": (c_int, [P]),
    "pevit_debug_occupy": (c_int, [P, c_int, c_int, C.c_double]),
    "pevit_op_lowrank_u": (c_int, [P, P, c_int, P, P, P, c_int, c_int, c_int, c_int]),
    "pevit_op_lowrank_grad": (c_int, [P, P, c_int, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int]),
    "pevit_op_lowrank_chunks": (c_int, [c_int]),
    "pevit_op_tn_chunks": (c_int, [c_int]),
    "pevit_op_tn_gemm64": (c_int, [P, P, c_int, P, c_int, P, P, P, c_int, c_int]),
    "pevit_op_lna_blocks": (c_int, [c_int]),
    "pevit_op_ln_bwd_affine": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int]),
    "pevit_op_colsum_reduce": (c_int, [P, P, c_int, c_int, P, P, P]),
    "pevit_op_prep_bottleneck": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, c_int]),
    "pevit_op_chain_bottleneck": (c_int, [P, c_int, P, P, P, P, P, c_int, c_size_t, c_size_t, c_size_t, c_size_t]),
    "pevit_op_im2col": (c_int, [P, P, P, c_int, c_int, c_int, c_int]),
    "pevit_op_im2col_u8": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int]),
    "pevit_tune": (c_int, [P, c_char_p, c_int]),
    "pevit_ar_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_size_t]),
    "pevit_ar_destroy": (None, [P]),
    "pevit_ar_handle_bytes": (c_int, []),
    "pevit_ar_export": (c_int, [P, P]),
    "pevit_ar_import": (c_int, [P, c_int, P]),
    "pevit_allreduce_flat": (c_int, [P, P, P, c_size_t]),
    "pevit_ar_error": (c_int, [P, P]),
    "pevit_ar_error_word": (c_void_p, [P]),
    "pevit_ar_fine_grained": (c_int, [P]),
    "pevit_ar_reset": (c_int, [P, P]),
    "pevit_set_external_poison": (c_int, [P, P]),
    "pevit_set_step_gate": (c_int, [P, P]),               # round 5: the fused step waits for this event behind its stem (pipelined DP)
    "pevit_streamk_error": (c_int, [P, P]),
    "pevit_streamk_status": (c_int, [P, P, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
}

_lib = None


class PevitError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach the signatures.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PevitError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    # torch first: its wheel carries its own libamdhip64; loaded AFTER this library (which would then pull the system runtime in)
    # the process holds two HIP runtimes, and the second one finds "no ROCm-capable device" (seen with `python __graft_entry__.py
    # smoke`, where build() loads the library before smoke() imports torch)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pevit_last_error()
        raise PevitError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
