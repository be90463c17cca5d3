"""ctypes signatures of include/ffno.h (single source of truth for the Python host AND the tests).

``bind(cdll)`` attaches argtypes/restype for every exported entry point and returns the list of
symbol names; a missing symbol raises AttributeError (the "exports every symbol" check).
"""
from __future__ import annotations

import ctypes as C

# generation of include/ffno.h these signatures and struct mirrors belong to (FFNO_ABI_VERSION there; _lib.check_abi compares
# it with what the loaded library reports before anything is called)
ABI_VERSION = 7
BRANCH_SELF_RANGE = 1      # ffno_fused_branch.flags: FFNO_BRANCH_SELF_RANGE (ffno_spectral_x3_mix_pair)

P = C.c_void_p
I = C.c_int
F = C.c_float
SZ = C.c_size_t
L = C.c_long


class WnDesc(C.Structure):
    """Mirror of ``ffno_wn_desc`` (include/ffno.h)."""
    _fields_ = [("g", P), ("v", P), ("w", P), ("dw", P), ("dg", P), ("dv", P),
                ("rows", C.c_int32), ("cols", C.c_int32)]


class FxPackDesc(C.Structure):
    """Mirror of ``ffno_fxpack_desc`` (include/ffno.h)."""
    _fields_ = [("src", P), ("dst", P), ("sh", C.c_int32), ("sc", C.c_int32), ("type", C.c_int32), ("pad_", C.c_int32)]


class FusedBranch(C.Structure):
    """Mirror of ``ffno_fused_branch`` (include/ffno.h)."""
    _fields_ = [("in_", P), ("out", P), ("resid", P), ("spec_save", P), ("planes", P), ("tw", P),
                ("B", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("axis", C.c_int32), ("accumulate", C.c_int32),
                ("planes_format", C.c_int32), ("tile_lines", C.c_int32), ("in_amax", P), ("out_amax", P),
                ("storage", C.c_int32), ("flags", C.c_int32), ("dft_frags", P)]


class FfOpts(C.Structure):
    """Mirror of ``ffno_ff_opts`` (include/ffno.h)."""
    _fields_ = [("in_amax", P), ("out_amax", P), ("max_workgroups", C.c_int32), ("schedule", C.c_int32),
                ("storage", C.c_int32)]


class LayerFwdDesc(C.Structure):
    """Mirror of ``ffno_layer_fwd_desc`` (include/ffno.h)."""
    _fields_ = [("a", FusedBranch), ("b", FusedBranch), ("branch_kernel", C.c_int32), ("interleave", C.c_int32),
                ("pk1", P), ("b1", P), ("pk2", P), ("b2", P), ("s_sum", P), ("resid", P), ("out", P), ("mask", P),
                ("P", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("ff_kernel", C.c_int32),
                ("ff_schedule", C.c_int32), ("ff_max_workgroups", C.c_int32), ("pad0_", C.c_int32), ("pad1_", C.c_int32), ("out_amax", P)]


class LayerBwdDesc(C.Structure):
    """Mirror of ``ffno_layer_bwd_desc`` (include/ffno.h)."""
    _fields_ = [("a", FusedBranch), ("b", FusedBranch), ("branch_kernel", C.c_int32), ("interleave", C.c_int32),
                ("g", P), ("g2", P), ("g_sum", P), ("mask", P), ("pk1b", P), ("pk2b", P), ("ds", P), ("s", P), ("pk1", P),
                ("b1", P), ("partial", P), ("nsplit", C.c_int32), ("P", C.c_int32), ("C", C.c_int32), ("H", C.c_int32),
                ("ff_kernel", C.c_int32), ("ff_schedule", C.c_int32), ("ff_max_workgroups", C.c_int32), ("pad_", C.c_int32), ("g_amax", P), ("s_amax", P), ("ds_amax", P)]


class LayerInferDesc(C.Structure):
    """Mirror of ``ffno_layer_infer_desc`` (include/ffno.h)."""
    _fields_ = [("a", FusedBranch), ("b", FusedBranch), ("interleave", C.c_int32), ("pad_", C.c_int32),
                ("pk1", P), ("b1", P), ("pk2", P), ("b2", P), ("resid", P), ("out", P), ("C", C.c_int32), ("H", C.c_int32),
                ("out_amax", P)]


class InferStackLayer(C.Structure):
    """Mirror of ``ffno_infer_stack_layer`` (include/ffno.h)."""
    _fields_ = [("planes_a", P), ("planes_b", P), ("pk1", P), ("b1", P), ("pk2", P), ("b2", P)]


class InferStackDesc(C.Structure):
    """Mirror of ``ffno_infer_stack_desc`` (include/ffno.h)."""
    _fields_ = [("a", FusedBranch), ("b", FusedBranch), ("layers", P), ("n_layers", C.c_int32), ("C", C.c_int32), ("H", C.c_int32),
                ("mode", C.c_int32), ("last_out", P), ("sync", P)]


class AmaxDesc(C.Structure):
    """Mirror of ``ffno_amax_desc`` (include/ffno.h)."""
    _fields_ = [("x", P), ("n", C.c_size_t)]


class FxRedDesc(C.Structure):
    """Mirror of ``ffno_fxred_desc`` (include/ffno.h)."""
    _fields_ = [("partial", P), ("dW1", P), ("dW2", P), ("db1", P), ("db2", P)]


class FfWgDesc(C.Structure):
    """Mirror of ``ffno_ffwg_desc`` (include/ffno.h)."""
    _fields_ = [("s", P), ("g", P), ("pk1", P), ("b1", P), ("pk1b", P), ("partial", P), ("s_amax", P), ("g_amax", P),
                ("s2", P), ("g2", P)]


class MarkovExtra(C.Structure):
    """Mirror of ``ffno_markov_extra`` (include/ffno.h)."""
    _fields_ = [("force", P), ("mu", P), ("use_position", C.c_int32), ("pad_", C.c_int32)]


class PadMap(C.Structure):
    """Mirror of ``ffno_padmap`` (include/ffno.h)."""
    _fields_ = [("size", C.c_int32 * 3), ("padded", C.c_int32 * 3)]


class PadDesc(C.Structure):
    """Mirror of ``ffno_pad_desc`` (include/ffno.h)."""
    _fields_ = [("plain", P), ("padded", P), ("R", C.c_int32), ("Cc", C.c_int32), ("inner", C.c_int32), ("Cp", C.c_int32)]


class FwPackDesc(C.Structure):
    """Mirror of ``ffno_fwpack_desc`` (include/ffno.h)."""
    _fields_ = [("w", P), ("wp", P), ("wpt", P), ("K", C.c_int32), ("real", C.c_int32)]


class X3PackDesc(C.Structure):
    """Mirror of ``ffno_x3pack_desc`` (include/ffno.h)."""
    _fields_ = [("planes", P), ("dst", P), ("K", C.c_int32), ("format", C.c_int32)]


class TrDesc(C.Structure):
    """Mirror of ``ffno_tr_desc`` (include/ffno.h)."""
    _fields_ = [("src", P), ("dst", P), ("rows", C.c_int32), ("cols", C.c_int32)]


SIGNATURES = {
    "ffno_build_target": (C.c_char_p, []),
    "ffno_abi_version": (I, []),
    "ffno_amax": (I, [P, SZ, P, P]),
    "ffno_amax_batched": (I, [P, I, SZ, P, P]),
    "ffno_lds_tr16_probe": (I, [P, I, P, P, P]),
    "ffno_twiddle_fill_host": (I, [P, I]),
    "ffno_dft_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "ffno_fw_pack": (I, [P, P, P, I, I, P]),
    "ffno_fw_pack_batched": (I, [P, I, I, I, P]),
    "ffno_mode_mix": (I, [P, P, P, I, I, I, I, P]),
    "ffno_dft_inv": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "ffno_fw_grad_partial": (I, [P, P, P, I, I, I, I, I, I, SZ, SZ, P]),
    "ffno_fw_grad_partial_multi": (I, [P, P, P, I, I, I, I, I, SZ, SZ, SZ, P]),
    "ffno_fw_grad_reduce_multi": (I, [P, P, I, I, I, I, SZ, I, I, P]),
    "ffno_fw_grad_partial_h2": (I, [P, P, P, I, I, I, I, I, I, SZ, SZ, P, P, I, P]),
    "ffno_fw_grad_partial_multi_h2": (I, [P, P, P, I, I, I, I, I, SZ, SZ, SZ, P, P, I, P]),
    "ffno_fw_grad_reduce": (I, [P, P, I, I, I, I, P]),
    "ffno_spectral_fused_pair": (I, [P, P, I, I, I, I, P]),
    "ffno_spectral_x3_supported": (I, [I, I, I]),
    "ffno_spectral_x3_pack_bytes": (SZ, [I, I]),
    "ffno_spectral_x3_dft_frags_bytes": (SZ, [I, I]),
    "ffno_spectral_x3_dft_frags": (I, [P, I, I, I, I, P, P]),
    "ffno_spectral_x3_pack": (I, [P, I, I, I, P]),
    "ffno_spectral_x3": (I, [P, I, I, I, I, P]),
    "ffno_spectral_x3_pair": (I, [P, P, I, I, I, I, I, P]),
    "ffno_spectral_x3_staged_supported": (I, [I, I, I]),
    "ffno_spectral_x3_staged_pair": (I, [P, P, P, P, I, I, I, I, P]),
    "ffno_layer_fwd": (I, [P, P]),
    "ffno_layer_bwd": (I, [P, P]),
    "ffno_infer_mix_bytes": (SZ, [I, I, I]),
    "ffno_layer_infer_supported": (I, [I, I, I, I, I, I, I]),
    "ffno_spectral_x3_mix_pair": (I, [P, P, I, I, P]),
    "ffno_infer_ff": (I, [P, P, P, P, P, P, P, P, I, I, P, P]),
    "ffno_infer_sum": (I, [P, P, P, I, P, P]),
    "ffno_layer_infer": (I, [P, P]),
    "ffno_infer_stack_supported": (I, [I, I, I, I, I, I, I, I]),
    "ffno_infer_stack_sync_words": (SZ, [I]),
    "ffno_infer_stack_trace_words": (SZ, [I]),
    "ffno_infer_stack": (I, [P, P]),
    "ffno_spectral_staged_pair": (I, [P, P, P, P, I, I, I, I, P]),
    "ffno_spectral_fused_supported": (I, [I, I, I]),
    "ffno_spectral_fused": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, P]),
    "ffno_spectral2d_ws_floats": (SZ, [I, I, I, I, I]),
    "ffno_spectral2d_path": (I, [I, I, I, I, I]),
    "ffno_spectral2d_weights_version": (I, [P, C.c_ulonglong]),
    "ffno_spectral2d_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "ffno_spectral2d_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "ffno_ff_mask_words": (SZ, [I, I]),
    "ffno_ff_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "ffno_ff_bwd_data": (I, [P, P, P, P, P, P, I, I, I, P]),
    "ffno_ff_wgrad_partial_floats": (SZ, [I, I, I]),
    "ffno_ff_bwd_weights_partial": (I, [P, P, P, P, P, I, I, I, I, P]),
    "ffno_ff_bwd_weights_reduce": (I, [P, P, P, P, P, I, I, I, I, P]),
    "ffno_ffx_supported": (I, [I, I]),
    "ffno_ffx_pack_bytes": (SZ, [I, I]),
    "ffno_ffx_pack": (I, [P, I, I, I, P]),
    "ffno_ffx_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "ffno_ffx_mask_unpack": (I, [P, P, I, I, I, P]),
    "ffno_ffx_bwd_data": (I, [P, P, P, P, P, I, I, I, P]),
    "ffno_ffx_fwd2": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "ffno_ffx_bwd_data2": (I, [P, P, P, P, P, P, P, I, I, I, P, P]),
    "ffno_ffx_bwd_weights_partial": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "ffno_ffx_bwd_weights_reduce": (I, [P, P, P, P, P, I, I, I, I, P]),
    "ffno_ffx_bwd_weights_reduce_batched": (I, [P, I, I, I, I, P]),
    "ffno_ffh_pack_bytes": (SZ, [I, I]),
    "ffno_ffh_pack": (I, [P, I, I, I, P]),
    "ffno_ffh_fwd2": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
    "ffno_ffh_bwd_data2": (I, [P, P, P, P, P, P, P, I, I, I, P, P]),
    "ffno_ffh_bwd_weights_partial": (I, [P, P, P, P, P, P, I, I, I, I, P, P, I, P]),
    "ffno_ffh_bwd_weights_partial_multi": (I, [P, I, I, I, I, I, I, I, P]),
    "ffno_layernorm_fwd": (I, [P, P, P, P, P, P, L, I, F, P]),
    "ffno_layernorm_nsplit": (I, [L]),
    "ffno_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L, I, I, P]),
    "ffno_weightnorm_fwd": (I, [P, I, I, P]),
    "ffno_weightnorm_bwd": (I, [P, I, I, P]),
    "ffno_transpose_batched": (I, [P, I, I, I, P]),
    "ffno_lift_fwd": (I, [P, P, P, P, I, I, I, P, P, P]),
    "ffno_lift_bwd": (I, [P, P, P, P, P, I, I, I, I, I, P, P]),
    "ffno_lift_bwd2": (I, [P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "ffno_lift_bwd_data": (I, [P, P, P, I, I, I, P, P]),
    "ffno_lift_fwd_bf16": (I, [P, P, P, P, I, I, I, P, P, P]),
    "ffno_lift_bwd_bf16": (I, [P, P, P, P, P, I, I, I, I, I, P, P]),
    "ffno_head_fwd_bf16": (I, [P, P, P, I, I, I, I, P, P]),
    "ffno_head_bwd_bf16": (I, [P, P, P, P, P, P, I, I, I, I, P, P, P]),
    "ffno_head_fold": (I, [P, P, P, P, P, I, I, I, P]),
    "ffno_head_fwd": (I, [P, P, P, I, I, I, I, P, P]),
    "ffno_head_bwd": (I, [P, P, P, P, P, P, I, I, I, I, P, P, P]),
    "ffno_head_param_grads": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "ffno_cdft_rows": (I, [P, P, I, I, I, I, I, P]),
    "ffno_fw2d_pack": (I, [P, P, P, P, I, I, P]),
    "ffno_fw2d_grad_reduce": (I, [P, P, P, I, I, I, I, P]),
    "ffno_cdft_rows2": (I, [P, P, I, I, I, I, I, I, P]),
    "ffno_fw3d_pack": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "ffno_fw3d_grad_reduce": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "ffno_dct_branch": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "ffno_fw_pack_real": (I, [P, P, P, I, I, P]),
    "ffno_mode_mix_real": (I, [P, P, P, I, I, I, P]),
    "ffno_fw_grad_reduce_real": (I, [P, P, I, I, I, I, P]),
    "ffno_cdft_rows_ws_floats": (SZ, [I, I, I, I]),
    "ffno_cdft_rows_mfma": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "ffno_fw2d_pack2": (I, [P, P, P, P, I, I, I, P]),
    "ffno_fw2d_grad_reduce2": (I, [P, P, P, I, I, I, I, I, P]),
    "ffno_plin_supported": (I, [I, I]),
    "ffno_plin_fwd": (I, [P, I, P, P, P, P, I, P, P, P, L, I, I, I, P]),
    "ffno_plin_bwd_data": (I, [P, I, P, P, P, I, P, L, I, I, I, I, P]),
    "ffno_plin_wgrad_nsplit": (I, [L]),
    "ffno_plin_wgrad_partial_floats": (SZ, [L, I, I]),
    "ffno_plin_bwd_weights": (I, [P, I, P, P, I, P, P, P, L, I, I, I, I, P]),
    "ffno_glin_supported": (I, [I, I]),
    "ffno_glin_fwd": (I, [P, P, P, P, P, L, I, I, I, F, C.c_uint32, P]),
    "ffno_glin_bwd_data": (I, [P, P, P, P, L, I, I, F, C.c_uint32, I, P]),
    "ffno_glin_wgrad_nsplit": (I, [L]),
    "ffno_glin_wgrad_partial_floats": (SZ, [L, I, I]),
    "ffno_glin_bwd_weights": (I, [P, P, P, P, P, P, L, I, I, F, C.c_uint32, I, P]),
    "ffno_dropout": (I, [P, SZ, F, C.c_uint32, P]),
    "ffno_dropout_mask": (I, [P, SZ, F, C.c_uint32, P]),
    "ffno_pad_copy": (I, [P, I, I, P]),
    "ffno_velocity_ws_floats": (SZ, [I, I, I]),
    "ffno_velocity_features": (I, [P, P, P, I, I, I, F, F, P]),
    "ffno_lploss_tmp_floats": (SZ, [I, I]),
    "ffno_lploss_fwd_bwd": (I, [P, P, P, P, P, I, I, F, P, P]),
    "ffno_markov_features": (I, [P, P, P, P, P, P, I, I, I, I, F, F, F, F, I, I, P, P]),
    "ffno_adamw_flat": (I, [P, P, P, P, SZ, F, F, F, F, F, I, F, P]),
    "ffno_adam_flat": (I, [P, P, P, P, SZ, F, F, F, F, F, I, F, P]),
    "ffno_axpy": (I, [P, P, F, SZ, P]),
}


def bind(lib: C.CDLL):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return list(SIGNATURES)


ERRORS = {-1: "FFNO_EINVAL (null pointer / bad size)", -2: "FFNO_EUNSUPPORTED (shape outside the compiled set)",
          -3: "FFNO_EMODES (modes > L/2+1)"}


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -3:
        raise ValueError(f"{what}: {ERRORS[rc]}")
    if rc in (-1, -2):
        raise ValueError(f"{what}: {ERRORS[rc]}")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t {rc}")
