"""ctypes binding of libu3d_hip.so (C ABI declared in include/u3d_hip.h).

PyTorch is used only as the owner of device memory and streams: every call passes raw `data_ptr()`s and the
current HIP stream.  There is NO CPU fallback: if the library is missing or a call fails, we raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("U3D_LIB_PATH") or os.path.join(_HERE, "libu3d_hip.so")      # override: kernel experiments (tools/build_variant.sh)

F32, BF16 = 0, 1


class U3DError(RuntimeError):
    pass


class BitGridStruct(C.Structure):
    _fields_ = [("words", C.c_void_p), ("prefix", C.c_void_p), ("batch", C.c_int32), ("dz", C.c_int32),
                ("dy", C.c_int32), ("dx", C.c_int32), ("layout", C.c_int32), ("row_capacity", C.c_int32)]


DL_NLIN, DL_NLN = 22, 7
(DL_RPH0, DL_RPH1, DL_RPH2, DL_QS0, DL_QS1, DL_QS2, DL_INQK, DL_INV, DL_OUTP, DL_OPROJ, DL_PE1, DL_FFN0, DL_FFN1, DL_REG0, DL_REG1,
 DL_REG2, DL_CLS0, DL_CLS1, DL_CLS2, DL_IOU0, DL_IOU1, DL_IOU2) = range(DL_NLIN)
(DLN_1, DLN_2, DLN_3, DLN_PE0, DLN_PE1, DLN_C1, DLN_C2) = range(DL_NLN)
DS_NAMES = ("SINE RPH1 RPH2 RAW QS1 QS2 QS POS QKIN QK V LSE O U1 MR QP SAMP GATED PEH0 UPE1 U2 X2C FFH U3 R1 R2 I1 I2 UC1 C1 UC2 "
            "C2 AMASK").split()
DG_NAMES = ("CLSO C2U C1U IOUO I2 I1 REGO R2 R1 F FFH OUT UPE1 P0 WL O2 DO DQK DV QS QS2 QS1 RAW RPH2 RPH1 LNP DU1 DPOSA "
            "SINE").split()


class DecLayerParams(C.Structure):
    """u3d_declayer_params (include/u3d_hip.h)."""
    _fields_ = [("w", C.c_void_p * DL_NLIN), ("wt", C.c_void_p * DL_NLIN), ("b", C.c_void_p * DL_NLIN),
                ("ln_g", C.c_void_p * DL_NLN), ("ln_b", C.c_void_p * DL_NLN), ("attw_w", C.c_void_p), ("attw_b", C.c_void_p),
                ("pe0_w", C.c_void_p), ("pe0_b", C.c_void_p), ("dim_t", C.c_void_p)]


class DecLayerDims(C.Structure):
    """u3d_declayer_dims."""
    _fields_ = [(n, C.c_int32) for n in ("m", "nq", "qps", "batch", "dz", "dy", "dx", "ncls", "code", "has_qs", "need_dref", "layer")] + \
               [("p_attn", C.c_float), ("p_drop", C.c_float), ("ln_eps", C.c_float), ("dtype", C.c_int32), ("dvalue_bf16", C.c_int32)]


class WPackDesc(C.Structure):
    """u3d_wpack_desc."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32),
                ("n_pad", C.c_int32), ("n_pad_t", C.c_int32), ("t_plain", C.c_int32), ("reserved", C.c_int32)]


_lib = None
_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_I3 = C.c_int32 * 3
_F3 = C.c_float * 3
_F6 = C.c_float * 6

_SIGS = {
    # name: (restype, argtypes)
    "u3d_version": (_I, []),
    "u3d_strerror": (C.c_char_p, [_I]),
    "u3d_points_augment": (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "u3d_boxes_augment": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "u3d_points_range_filter": (_I, [_P, _P, _I, _I, C.POINTER(C.c_float), _P, _P, _P]),
    "u3d_point_sample": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "u3d_boxes_range_filter": (_I, [_P, _P, _P, _I, _I, C.POINTER(C.c_float), _P, _P]),
    "u3d_event_create": (_I, [C.POINTER(C.c_void_p)]),
    "u3d_event_record": (_I, [C.c_void_p, _I, _P]),
    "u3d_event_elapsed_ms": (_I, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "u3d_event_destroy": (_I, [C.c_void_p]),
    "u3d_bitgrid_nwords": (_L, [_I, _I, _I, _I]),
    "u3d_bitgrid_nwords_layout": (_L, [_I, _I, _I, _I, _I]),
    "u3d_voxelize_dynamic": (_I, [_P, _P, _I, _I, _I, _F3, _F6, _P, _P]),
    "u3d_scatter_mean": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "u3d_bitgrid_mark": (_I, [C.POINTER(BitGridStruct), _P, _I, _P]),
    "u3d_bitgrid_mark_strided": (_I, [C.POINTER(BitGridStruct), _P, _P, _I, _I3, _I3, _I3, _P]),
    "u3d_bitgrid_scan_scratch": (_L, [_L]),
    "u3d_bitgrid_scan": (_I, [C.POINTER(BitGridStruct), _P, _P]),
    "u3d_bitgrid_rank": (_I, [C.POINTER(BitGridStruct), _P, _I, _P, _P]),
    "u3d_bitgrid_coords": (_I, [C.POINTER(BitGridStruct), _P, _I, _P]),
    "u3d_nbr_table": (_I, [C.POINTER(BitGridStruct), _P, _P, _I, _I3, _I3, _I3, _I, _P, _I, _P]),
    "u3d_dense_nbr_table": (_I, [_I, _I3, _I3, _I3, _I3, _I3, _I, _P, _I, _P]),
    "u3d_voxelize_hard_workspace": (_L, [_I, _I, _I]),
    "u3d_voxelize_hard": (_I, [_P, _P, _I, _I, _I, _I, _F3, _F6, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P]),
    "u3d_spconv_fwd": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "u3d_spconv_wgrad_workspace": (_L, [_I, _I, _I, _I]),
    "u3d_spconv_wgrad": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "u3d_igemm_fwd_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "u3d_igemm_fwd_stats_blocks": (_I, [_I, _I, _I, _I]),
    "u3d_igemm_fwd_stats_rows": (_I, [_I, _I, _I, _I]),
    "u3d_igemm_fwd_add_bf16": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "u3d_linear_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "u3d_igemm_wgrad_bf16_workspace": (_L, [_I, _I, _I, _I]),
    "u3d_igemm_wgrad_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "u3d_colsum_workspace": (_L, [_I, _I]),
    "u3d_colsum": (_I, [_P, _I, _I, _I, _P, _P, _L, _P]),
    "u3d_bn_stats_workspace": (_L, [_I, _I]),
    "u3d_bn_stats": (_I, [_P, _P, _I, _I, _I, _P, _P, _L, _P]),
    "u3d_bn_finalize": (_I, [_P, _P, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    "u3d_bn_forward_stats": (_I, [_P, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _L, _P]),
    "u3d_bn_finalize_partials": (_I, [_P, _I, _I, _P, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    "u3d_igemm_fwd_stats_tile_rows": (_I, [_I, _I, _I]),
    "u3d_igemm_fwd_stats_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "u3d_igemm_fwd_split_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "u3d_split_rows_f32": (_I, [_P, _P, _I, _I, _P, _P]),
    "u3d_split3_weights": (_I, [_P, C.c_int64, C.c_int64, C.c_int64, _I, _I, _I, _P, _P]),
    "u3d_sum3_f32": (_I, [_P, _P, _P, _P, C.c_int64, _P]),
    "u3d_split_rows_batch": (_I, [_P, _P]),
    "u3d_split3_job_bytes": (C.c_int64, []),
    "u3d_split3_job_blocks": (_I, [_I, _I, _I]),
    "u3d_split3_weights_batch": (_I, [_P, _I, _I, _P]),
    "u3d_igemm_direct_split_bf16": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "u3d_igemm_dgrad_bnstats_bf16": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "u3d_bn_bwd_finalize_partials": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P]),
    "u3d_subm_halo_sizes": (_I, [_I, _P, _P, _P]),
    "u3d_subm_halo_build": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P]),
    "u3d_subm_halo_wpack128": (_I, [_P, _P, _I, _P]),
    "u3d_subm_halo_wpack128_batched": (_I, [_P, _P, _I, _I, _P]),
    "u3d_subm_halo_conv128_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    "u3d_subm_halo_wgrad64_workspace": (_L, []),
    "u3d_subm_halo_wgrad64_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _P]),
    "u3d_subm_halo_wpack": (_I, [_P, _P, _P]),
    "u3d_subm_halo_wpack_batched": (_I, [_P, _P, _I, _P]),
    "u3d_subm_halo_conv64_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P]),
    "u3d_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "u3d_bn_bwd_stats": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P]),
    "u3d_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    "u3d_bn_apply_planes": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "u3d_bn_bwd_apply_planes": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    "u3d_to_dense": (_I, [_P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P]),
    "u3d_from_dense": (_I, [_P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P]),
    "u3d_fps": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _L, _P, _L, _I, _P]),
    "u3d_refine_decode_fwd": (_I, [_P, _P, _P, _I, _I, _P, C.c_float, _P, _P, _P, _P]),
    "u3d_loss_targets": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "u3d_fps2": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _L, _P, _L, _I, _P]),
    "u3d_fps_prep": (_I, [_P, _I, _P, _P, _I, _I, _P, _P, _P, _P]),
    "u3d_fps_points": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    "u3d_query_embed_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "u3d_query_embed_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "u3d_match_cost": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "u3d_lsa": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "u3d_trilinear_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "u3d_trilinear_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "u3d_nms3d_workspace": (_L, [_I]),
    "u3d_nms3d": (_I, [_P, _P, _I, C.c_float, _P, _P, _L, _P]),
    "u3d_iou3d_rotated_aligned": (_I, [_P, _P, _I, _P, _P]),
    "u3d_tap_gather_sum": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "u3d_tap_gather_sum_add": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "u3d_wgrad_batched_workspace": (_L, [_I, _I, _I, _I]),
    "u3d_wgrad_batched_bf16": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _L, _P]),
    "u3d_skinny_wgrad_chunks": (_I, [_I]),
    "u3d_skinny_wgrad_bf16": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "u3d_skinny_wgrad_batched": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "u3d_colsum_batched_workspace": (_L, [_I, _I, _I]),
    "u3d_colsum_batched": (_I, [_P, _P, _I, _I, _I, _I, _P, _L, _P]),
    "u3d_layernorm_blocks": (_I, [_I]),
    "u3d_layernorm_fwd": (_I, [_P, _I, _I, _I, _P, _P, C.c_float, _I, _P, _I, _P, _P, _P]),
    "u3d_layernorm_bwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "u3d_det_loss_workspace": (_L, [_I, _I]),
    "u3d_det_loss_fwd": (_I, [_P] * 10 + [_I] * 5 + [C.c_float] * 5 + [_P, _P, _L, _P]),
    "u3d_det_loss_bwd": (_I, [_P] * 11 + [_I] * 5 + [C.c_float] * 5 + [_P, _P, _P, _P]),
    "u3d_denormalize_boxes": (_I, [_P, _I, _I, _P, _P]),
    "u3d_box_decode_fwd": (_I, [_P, _I, _P, _I, _I, _F6, C.c_float, _P, _P]),
    "u3d_box_decode_bwd": (_I, [_P, _I, _P, _P, _I, _I, _F6, C.c_float, _P, _P, _P]),
    "u3d_sine_embed_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "u3d_sine_embed_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "u3d_cast_bf16": (_I, [_P, _P, _L, _P]),
    "u3d_permute_block_elems": (_I, []),
    "u3d_permute_bf16_tiled": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "u3d_permute_bf16_batched": (_I, [_P, _P, _P, _P, _I, _P]),
    "u3d_adamw_workspace": (_L, [_L]),
    "u3d_adamw_step": (_I, [_P, _P, _P, _P, _L, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _L, _P]),
    "u3d_adamw_set_hyper": (_I, [_P] + [C.c_float] * 6 + [_P]),
    "u3d_adamw_step_state": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _L, _P]),
    "u3d_adamw_step_hold": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P]),
    "u3d_capacity_flag": (_I, [_P, _P, _I, _P, _P]),
    "u3d_gather_rows": (_I, [_P, _P, _I, _I, _P, _P]),
    "u3d_soft_nms": (_I, [_P, _P, _P, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P]),
    "u3d_box_merge_workspace": (_L, [_I]),
    "u3d_box_merge": (_I, [_P, _P, _I, C.c_float, _P, _P, _P, _L, _P]),
    "u3d_decoder_layer_slots": (_I, [_I, _I, _I, _P, _P]),
    "u3d_decoder_layer_blocks": (_I, [_I]),
    "u3d_decoder_layer_fwd": (_I, [C.POINTER(DecLayerParams), C.POINTER(DecLayerDims)] + [_P] * 11 + [_L, _P]),
    "u3d_decoder_layer_bwd": (_I, [C.POINTER(DecLayerParams), C.POINTER(DecLayerDims)] + [_P] * 15 + [_L, _P]),
    "u3d_mha_fwd": (_I, [_P, _P, _I, _I, C.c_float, _I, _P, _P, _P, _P]),
    "u3d_mha_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _I, _P, _P, _P, _P]),
    "u3d_wpack_bf16": (_I, [_P, _I, _I, _P]),
    "u3d_wpack": (_I, [_P, _I, _I, _I, _P]),
    "u3d_decoder_layer_slots_dt": (_I, [_I, _I, _I, _I, _P, _P]),
    "u3d_decoder_layer_blocks_dt": (_I, [_I, _I]),
    "u3d_mha_fwd_dt": (_I, [_P, _P, _I, _I, C.c_float, _I, _P, _P, _P, _I, _P]),
    "u3d_mha_bwd_dt": (_I, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _I, _P, _P, _P, _I, _P]),
    "u3d_dropout_mask": (_I, [_P, _I, _I, _L, C.c_float, _I, _P, _P]),
    "u3d_scatter_rows": (_I, [_P, _P, _I, _I, _P, _P]),
}


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise U3DError(f"{LIB_PATH} is missing: run `python -m uni3detr_amd.build` (hipcc, gfx950). "
                           "There is no CPU/PyTorch fallback for the HIP hot path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)   # AttributeError -> loud failure if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise U3DError(f"{what} failed: {lib().u3d_strerror(rc).decode()} ({rc})")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise U3DError(f"unsupported dtype {t.dtype}")


# --------------------------------------------------------------------------------------------------
# BitGrid
# --------------------------------------------------------------------------------------------------
class BitGrid:
    """Occupancy lattice of one sparse level (see include/u3d_hip.h)."""

    def __init__(self, batch, dims, device, linear=False):
        self.batch, self.dims = int(batch), tuple(int(d) for d in dims)
        self.linear = bool(linear)
        self.nwords = int(lib().u3d_bitgrid_nwords_layout(self.batch, *self.dims, 1 if linear else 0))
        self.words = torch.zeros(self.nwords, dtype=torch.int64, device=device)
        self.prefix = torch.empty(self.nwords + 1, dtype=torch.int32, device=device)
        self._scratch = torch.empty(int(lib().u3d_bitgrid_scan_scratch(self.nwords)), dtype=torch.int32, device=device)
        self.c = BitGridStruct(self.words.data_ptr(), self.prefix.data_ptr(), self.batch, *self.dims, 1 if linear else 0, 0)

    def set_row_capacity(self, cap):
        self.c.row_capacity = int(cap)

    def mark(self, coors):
        _check(lib().u3d_bitgrid_mark(C.byref(self.c), _ptr(coors), coors.shape[0], _stream()), "bitgrid_mark")

    def mark_strided(self, in_coors, n_dev, ksize, stride, pad):
        _check(lib().u3d_bitgrid_mark_strided(C.byref(self.c), _ptr(in_coors), _ptr(n_dev), in_coors.shape[0],
                                              _I3(*ksize), _I3(*stride), _I3(*pad), _stream()), "bitgrid_mark_strided")

    def scan(self):
        _check(lib().u3d_bitgrid_scan(C.byref(self.c), _ptr(self._scratch), _stream()), "bitgrid_scan")

    @property
    def count_dev(self):
        """int32 device scalar view: number of occupied cells (valid after scan())."""
        return self.prefix[self.nwords:]

    def rank(self, coors):
        out = torch.empty(coors.shape[0], dtype=torch.int32, device=coors.device)
        _check(lib().u3d_bitgrid_rank(C.byref(self.c), _ptr(coors), coors.shape[0], _ptr(out), _stream()), "bitgrid_rank")
        return out

    def coords(self, n):
        out = torch.empty((n, 4), dtype=torch.int32, device=self.words.device)      # (rows past the count: -1, written by the kernel)
        _check(lib().u3d_bitgrid_coords(C.byref(self.c), _ptr(out), n, _stream()), "bitgrid_coords")
        return out

    def nbr_table(self, q_coors, n_dev, ksize, stride, pad, mode):
        n = q_coors.shape[0]
        ld = (n + 127) // 128 * 128
        kvol = ksize[0] * ksize[1] * ksize[2]
        nbr = torch.empty((kvol, ld), dtype=torch.int32, device=q_coors.device)
        _check(lib().u3d_nbr_table(C.byref(self.c), _ptr(q_coors), _ptr(n_dev), n, _I3(*ksize), _I3(*stride), _I3(*pad),
                                   mode, _ptr(nbr), ld, _stream()), "nbr_table")
        return nbr


def dense_nbr_table(batch, q_dims, t_dims, ksize, stride, pad, mode, device):
    n = batch * q_dims[0] * q_dims[1] * q_dims[2]
    ld = (n + 127) // 128 * 128
    kvol = ksize[0] * ksize[1] * ksize[2]
    nbr = torch.empty((kvol, ld), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _check(lib().u3d_dense_nbr_table(batch, _I3(*q_dims), _I3(*t_dims), _I3(*ksize), _I3(*stride), _I3(*pad), mode, _ptr(nbr), ld,
                                         _stream()), "dense_nbr_table")
    return nbr


# --------------------------------------------------------------------------------------------------
# voxelization
# --------------------------------------------------------------------------------------------------
def voxelize_hard(points, scene_off, batch, max_pts_per_scene, voxel_size, pc_range, max_points, max_voxels,
                  want_voxels=True, want_mean=True):
    """points f32 [n_total,F] (scenes concatenated), scene_off int32 [B+1] device.
    Returns (voxels|None, coors, num_points, mean|None, voxel_off) with capacity B*max_voxels rows."""
    n_total, nfeat = points.shape
    cap = batch * max_voxels
    dev = points.device
    voxels = torch.empty((cap, max_points, nfeat), dtype=torch.float32, device=dev) if want_voxels else None
    coors = torch.full((cap, 4), -1, dtype=torch.int32, device=dev)     # rows past the voxel count stay (-1,..): inert everywhere
    num = torch.zeros((cap,), dtype=torch.int32, device=dev)
    mean = torch.zeros((cap, nfeat), dtype=torch.float32, device=dev) if want_mean else None
    voxel_off = torch.empty((batch + 1,), dtype=torch.int32, device=dev)
    wsb = int(lib().u3d_voxelize_hard_workspace(n_total, batch, max_pts_per_scene))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _check(lib().u3d_voxelize_hard(_ptr(points), _ptr(scene_off), batch, n_total, max_pts_per_scene, nfeat,
                                   _F3(*voxel_size), _F6(*pc_range), max_points, max_voxels, _ptr(voxels), _ptr(coors),
                                   _ptr(num), _ptr(mean), _ptr(voxel_off), _ptr(ws), wsb, _stream()), "voxelize_hard")
    return voxels, coors, num, mean, voxel_off


def voxelize_dynamic(points, scene_off, batch, voxel_size, pc_range):
    """-> coors int32 [n_total,4] (b,z,y,x), (b,-1,-1,-1) for out-of-range points."""
    n_total, nfeat = points.shape
    coors = torch.empty((n_total, 4), dtype=torch.int32, device=points.device)
    _check(lib().u3d_voxelize_dynamic(_ptr(points), _ptr(scene_off), batch, n_total, nfeat, _F3(*voxel_size), _F6(*pc_range),
                                      _ptr(coors), _stream()), "voxelize_dynamic")
    return coors


def scatter_mean(points, rank, n_voxels):
    n_total, nfeat = points.shape
    sums = torch.zeros((n_voxels, nfeat), dtype=torch.float32, device=points.device)
    counts = torch.zeros((n_voxels,), dtype=torch.int32, device=points.device)
    _check(lib().u3d_scatter_mean(_ptr(points), _ptr(rank), n_total, nfeat, _ptr(sums), _ptr(counts), n_voxels, _stream()), "scatter_mean")
    return sums, counts


# --------------------------------------------------------------------------------------------------
# sparse conv / BN / dense
# --------------------------------------------------------------------------------------------------
class ExternalEvent:
    """HIP event recorded with hipEventRecordExternal on torch's current stream (torch.cuda.Event(external=True) is refused on
    ROCm builds): inside a capture it becomes an event-record node of the graph."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().u3d_event_create(C.byref(h)), "event_create")
        self.h = h

    def record(self):
        _check(lib().u3d_event_record(self.h, 1, _stream()), "event_record")

    def elapsed_time(self, other):
        ms = C.c_float()
        _check(lib().u3d_event_elapsed_ms(self.h, other.h, C.byref(ms)), "event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            lib().u3d_event_destroy(self.h)
        except Exception:
            pass


class KernelTimer:
    """HIP-event timing of individual launches on the stream they run on (used by bench.py for the roofline block).
    mode 'census': also counts the valid rulebook pairs P of each call (host sync) to price its algorithmic bytes:
    N_in*Cin*s + N_out*Cout*s + 8*P + K*Cin*Cout*s."""

    def __init__(self, mode="time", targets=(), per_step=0):
        """mode 'mark': only the calls whose index within the step (call counter modulo `per_step`) is in `targets` get events,
        and those are EXTERNAL events (hipEventRecordExternal): recorded inside a hipGraph capture they become event-record nodes
        of the graph, so after a replay `marks[i]` times that launch as it ran INSIDE the replayed step."""
        self.mode, self.calls, self.census = mode, [], []
        self.targets, self.per_step, self.counter, self.marks = set(targets), per_step, 0, {}

    def begin(self):
        if self.mode == "off":
            return None
        if self.mode == "mark":
            i = self.counter % self.per_step if self.per_step else self.counter
            self.counter += 1
            if i not in self.targets:
                return None
            try:
                e = ExternalEvent()
                e.record()
            except U3DError:                  # no event-record nodes on this runtime: keep the capture alive, time nothing
                self.mode, self.marks = "off", {}
                return None
            return (i, e)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, e0, meta=None):
        if self.mode == "off":
            return
        if self.mode == "mark":
            if e0 is not None:
                try:
                    e1 = ExternalEvent()
                    e1.record()
                    self.marks[e0[0]] = (tag, e0[1], e1)      # the latest recording wins: the captured one
                except U3DError:
                    self.mode, self.marks = "off", {}
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.calls.append((tag, e0, e1))
        if self.mode == "census":
            self.census.append((tag, meta))

    def mark_durations_ms(self):
        """{index: ms} of the marked launches in the most recent (replayed) step."""
        torch.cuda.synchronize()
        return {i: a.elapsed_time(b) for i, (t, a, b) in self.marks.items()}

    def durations_ms(self):
        torch.cuda.synchronize()
        return [(t, a.elapsed_time(b)) for t, a, b in self.calls]


TIMER = None
CALL_KIND = "sparse"
USE_IGEMM_V2 = True


def spconv_fwd_stats(inp, w_nmajor, nbr, n_out_dev, n_out, cout):
    """Forward with n-major weights [K, Cout, Cin] + per-row-tile BatchNorm statistics of the output.
    -> (out [n_out, cout], stats f64 [nblocks, 2, cout], tile_rows) or None when the shape is not served by that kernel."""
    cin, kvol = inp.shape[1], w_nmajor.shape[0]
    if inp.dtype != torch.bfloat16 or not USE_IGEMM_V2:
        return None
    nblocks = int(lib().u3d_igemm_fwd_stats_blocks(n_out, cin, cout, kvol)) if (nbr is not None or kvol == 1) else 0
    if nblocks == 0:
        return None
    # 0 for the direct-operand kernels of the narrow levels: per-wave partials, all of them count (rows_per_block = 0 downstream)
    tr = int(lib().u3d_igemm_fwd_stats_rows(n_out, cin, cout, kvol)) if nbr is not None else int(lib().u3d_igemm_fwd_stats_tile_rows(n_out, cin, cout))
    if tr and nblocks != (n_out + tr - 1) // tr:
        tr = 0
    out = torch.empty((n_out, cout), dtype=inp.dtype, device=inp.device)
    stats = torch.empty((nblocks, 2, cout), dtype=torch.float64, device=inp.device)
    ld = nbr.shape[1] if nbr is not None else 0
    t = TIMER
    e0 = t.begin() if t is not None else None
    _check(lib().u3d_igemm_fwd_stats_bf16(_ptr(inp), _ptr(w_nmajor), _ptr(nbr), ld, _ptr(out), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                          _ptr(stats), _stream()), "igemm_fwd_stats_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            pairs = int((nbr[:, :n_out] >= 0).sum().item()) if nbr is not None else n_out
            meta = dict(kind=CALL_KIND, v2=True, n_in=inp.shape[0], n_out=n_out, cin=cin, cout=cout, kvol=kvol, pairs=pairs,
                        bytes=inp.shape[0] * cin * 2 + n_out * cout * 2 + 8 * pairs + kvol * cin * cout * 2, flops=2 * pairs * cin * cout)
        t.end("spconv_fwd", e0, meta)
    return out, stats, tr


class BnEpi(C.Structure):
    """u3d_bn_epi (include/u3d_hip.h): the BatchNorm whose dy an input-gradient launch writes."""
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("relu", C.c_int32), ("reserved", C.c_int32)]

    @staticmethod
    def of(x, y, mean, invstd, gamma, beta, relu):
        p = lambda t: None if t is None else t.data_ptr()        # noqa: E731
        return BnEpi(p(x), p(y), p(mean), p(invstd), p(gamma), p(beta), int(relu), 0)


def spconv_dgrad_bnstats(dout, w_nmajor, nbr, n_dev, n, cout, addend, epi):
    """Input gradient (weights [K, Cout_gemm, Cin_gemm] n-major as the dgrad passes them) + addend + per-row-tile BatchNorm-backward
    sums of the layer that produced the conv's input (u3d_igemm_dgrad_bnstats_bf16).  -> (din, partial f64 [tiles, 2, cout],
    tile_rows) or None when no kernel with that epilogue serves the shape."""
    kvol, cin = w_nmajor.shape[0], dout.shape[1]
    if dout.dtype != torch.bfloat16 or not USE_IGEMM_V2 or nbr is None:
        return None
    tr = int(lib().u3d_igemm_fwd_stats_rows(n, cin, cout, kvol))
    if tr == 0:
        return None
    out = torch.empty((n, cout), dtype=dout.dtype, device=dout.device)
    stats = torch.empty(((n + tr - 1) // tr, 2, cout), dtype=torch.float64, device=dout.device)
    nbr_p, ld = _nbr_ptr_ld(nbr)
    t = TIMER
    e0 = t.begin() if t is not None else None
    rc = lib().u3d_igemm_dgrad_bnstats_bf16(_ptr(dout), _ptr(w_nmajor), nbr_p, ld, _ptr(addend), _ptr(out), _ptr(n_dev), n, cin, cout, kvol,
                                            C.byref(epi), _ptr(stats), _stream())
    if rc == -2:
        return None
    _check(rc, "igemm_dgrad_bnstats_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            tb = nbr.t.flip(0) if isinstance(nbr, RevNbr) else nbr
            pairs = int((tb[:, :n] >= 0).sum().item())
            meta = dict(kind=CALL_KIND, v2=True, n_in=dout.shape[0], n_out=n, cin=cin, cout=cout, kvol=kvol, pairs=pairs,
                        bytes=dout.shape[0] * cin * 2 + n * cout * 2 + 8 * pairs + kvol * cin * cout * 2, flops=2 * pairs * cin * cout)
        t.end("spconv_dgrad", e0, meta)
    return out, stats, tr


def bn_bwd_finalize_partials(partial, tile_rows, n_dev, n_cap):
    """per-row-tile (sum g, sum g * xhat) -> what bn_bwd_stats(..., want_f32=True) returns."""
    nb, _, c = partial.shape
    sums = torch.empty((2, c), dtype=torch.float64, device=partial.device)
    s32 = torch.empty((2, c), dtype=torch.float32, device=partial.device)
    _check(lib().u3d_bn_bwd_finalize_partials(_ptr(partial), nb, tile_rows, _ptr(n_dev), n_cap, c, _ptr(sums), _ptr(s32), _stream()),
           "bn_bwd_finalize_partials")
    return sums, s32


class SubmHalo:
    """Per-tile distinct-row lists + 16-bit slot tables of one SubM level (u3d_subm_halo_build): built once per level and step from
    the forward neighbour table, shared by all of the level's 64 -> 64 convs and (offsets reversed) their input gradients."""
    TILE = 128

    MAX_ROWS = 140 * 256 * 32       # u3d_subm_halo_build's LDS row bitmap: 4.5 B per 32 rows + 1 KiB within 160 KiB (the C side decides)

    def __init__(self, nbr_fwd, n_dev, n_cap):
        dev = nbr_fwd.device
        a, b, t = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _check(lib().u3d_subm_halo_sizes(n_cap, C.byref(a), C.byref(b), C.byref(t)), "subm_halo_sizes")
        self.tiles, self.n_dev, self.n_cap, self.nbr = t.value, n_dev, n_cap, nbr_fwd
        self.kvol = nbr_fwd.shape[0]                     # <= 27 offsets (27: sparse SubM levels; 9: the dense stack's (1,3,3) convs)
        self.tile_rows = torch.empty((a.value,), dtype=torch.int32, device=dev)
        self.loc = torch.empty((b.value,), dtype=torch.int16, device=dev)
        self.tile_cnt = torch.empty((t.value,), dtype=torch.int32, device=dev)
        rc = lib().u3d_subm_halo_build(_ptr(nbr_fwd), nbr_fwd.shape[1], _ptr(n_dev), n_cap, _ptr(self.tile_rows), _ptr(self.loc),
                                       _ptr(self.tile_cnt), self.kvol, _stream())
        self.ok = rc != -2              # U3D_ERR_UNSUPPORTED: more rows than the LDS row bitmap holds - the caller keeps the table kernels
        if self.ok:
            _check(rc, "subm_halo_build")


def subm_halo_wgrad(x, dy, halo, out=None, max_slots=0):
    """dW f32 [27, 64, 64] of a 64 -> 64 SubM conv from the level's halo tables (u3d_subm_halo_wgrad64_bf16); out: contiguous f32
    tensor of 27 * 64 * 64 elements to write into."""
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.shape == dy.shape == (halo.n_cap, 64)
    dw = out.view(27, 64, 64) if (out is not None and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == 27 * 4096) \
        else torch.empty((27, 64, 64), dtype=torch.float32, device=x.device)
    wsb = int(lib().u3d_subm_halo_wgrad64_workspace())
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    t = TIMER
    e0 = t.begin() if t is not None else None
    _check(lib().u3d_subm_halo_wgrad64_bf16(_ptr(x), _ptr(dy), _ptr(halo.tile_rows), _ptr(halo.loc), _ptr(halo.tile_cnt), _ptr(halo.n_dev),
                                            halo.n_cap, _ptr(dw), _ptr(ws), wsb, int(max_slots), _stream()), "subm_halo_wgrad64_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            n = halo.n_cap
            pairs = int((halo.nbr[:, :n] >= 0).sum().item())
            meta = dict(kind=CALL_KIND, v2=True, n_in=n, n_out=n, cin=64, cout=64, kvol=27, pairs=pairs,
                        bytes=n * 64 * 2 * 2 + 8 * pairs + 27 * 64 * 64 * 4, flops=2 * pairs * 64 * 64)
        t.end("spconv_wgrad", e0, meta)
    return dw


def subm_halo_wpack(w_nmajor, out=None):
    """bf16 [27, 64 (out), 64 (reduction)] -> the MFMA fragment order u3d_subm_halo_conv64_bf16 reads (same shape and size)."""
    k, c = w_nmajor.shape[0], w_nmajor.shape[1]
    assert w_nmajor.dtype == torch.bfloat16 and w_nmajor.is_contiguous() and w_nmajor.shape[2] == c
    assert (k == 27 and c == 64) or (1 <= k <= 27 and c == 128)
    out = torch.empty_like(w_nmajor) if out is None else out
    if c == 64:
        _check(lib().u3d_subm_halo_wpack(_ptr(w_nmajor), _ptr(out), _stream()), "subm_halo_wpack")
    else:
        _check(lib().u3d_subm_halo_wpack128(_ptr(w_nmajor), _ptr(out), k, _stream()), "subm_halo_wpack128")
    return out


def subm_halo_wpack_plan(pairs, device):
    """[(src, dst)] of [27, C, C] bf16 tensors, all with the same C in (64, 128) -> plan for subm_halo_wpack_batched (device pointer
    arrays; the tensors must stay alive)."""
    k, c = pairs[0][0].shape[0], pairs[0][0].shape[1]
    assert all(tuple(a.shape) == (k, c, c) for a, _ in pairs) and ((k == 27 and c == 64) or c == 128)
    src = torch.tensor([a.data_ptr() for a, _ in pairs], dtype=torch.int64, device=device)
    dst = torch.tensor([b.data_ptr() for _, b in pairs], dtype=torch.int64, device=device)
    return src, dst, len(pairs), pairs, c, k


def subm_halo_wpack_batched(plan):
    if plan[4] == 64:
        _check(lib().u3d_subm_halo_wpack_batched(_ptr(plan[0]), _ptr(plan[1]), plan[2], _stream()), "subm_halo_wpack_batched")
    else:
        _check(lib().u3d_subm_halo_wpack128_batched(_ptr(plan[0]), _ptr(plan[1]), plan[2], plan[5], _stream()), "subm_halo_wpack128_batched")


def subm_halo_conv(inp, w_packed, halo, krev=False, addend=None, want_stats=False, tag="spconv_fwd", bn_epi=None, max_slots=0):
    """64 -> 64 channel, 27-offset SubM conv out of the tile's staged distinct rows (u3d_subm_halo_conv64_bf16).
    w_packed: subm_halo_wpack of bf16 [27, 64 (out), 64 (reduction)].  -> out, or (out, stats f64 [tiles, 2, 64], 128) with want_stats."""
    c = inp.shape[1]
    assert inp.dtype == torch.bfloat16 and c in (64, 128) and tuple(w_packed.shape) == (halo.kvol, c, c) and inp.shape[0] == halo.n_cap
    assert c == 128 or halo.kvol == 27
    assert c == 64 or bn_epi is None, "the BatchNorm-backward epilogue exists on the 64-channel kernel only"
    out = torch.empty_like(inp)
    stats = torch.empty((halo.tiles, 2, c), dtype=torch.float64, device=inp.device) if want_stats else None
    t = TIMER
    e0 = t.begin() if t is not None else None
    if c == 64:
        _check(lib().u3d_subm_halo_conv64_bf16(_ptr(inp), _ptr(w_packed), _ptr(halo.tile_rows), _ptr(halo.loc), _ptr(halo.tile_cnt),
                                               _ptr(halo.n_dev), halo.n_cap, int(krev), _ptr(addend), _ptr(out), _ptr(stats),
                                               None if bn_epi is None else C.byref(bn_epi), int(max_slots), _stream()),
               "subm_halo_conv64_bf16")
    else:
        _check(lib().u3d_subm_halo_conv128_bf16(_ptr(inp), _ptr(w_packed), _ptr(halo.tile_rows), _ptr(halo.loc), _ptr(halo.tile_cnt),
                                                _ptr(halo.n_dev), halo.n_cap, int(krev), _ptr(addend), _ptr(out), _ptr(stats),
                                                int(max_slots), halo.kvol, _stream()), "subm_halo_conv128_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            n = halo.n_cap
            pairs = int((halo.nbr[:, :n] >= 0).sum().item())
            meta = dict(kind=CALL_KIND, v2=True, n_in=n, n_out=n, cin=c, cout=c, kvol=halo.kvol, pairs=pairs,
                        bytes=n * c * 2 * 2 + 8 * pairs + halo.kvol * c * c * 2, flops=2 * pairs * c * c)
        t.end(tag, e0, meta)
    return (out, stats, SubmHalo.TILE) if want_stats else out


def bn_finalize_partials(stats, tile_rows, n_dev, n_cap, eps, momentum, running_mean=None, running_var=None, num_batches=None):
    nblocks, _, c = stats.shape
    mean = torch.empty((c,), dtype=torch.float32, device=stats.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=stats.device)
    _check(lib().u3d_bn_finalize_partials(_ptr(stats), nblocks, tile_rows, _ptr(n_dev), n_cap, c, eps, momentum, _ptr(running_mean),
                                          _ptr(running_var), _ptr(num_batches), _ptr(mean), _ptr(invstd), _stream()), "bn_finalize_partials")
    return mean, invstd


class RevNbr:
    """A neighbour table read with its offsets in REVERSED order (row K-1-k for offset k): what the transposed table of a
    submanifold convolution is - voxel j is the neighbour of m at offset +d exactly when m is the neighbour of j at -d, and the
    3x3x3 offsets are enumerated symmetrically.  Handed to the kernels as a pointer to the last table row with a negative row
    stride: the input gradient of a SubM layer needs no table of its own (4 table builds per step less)."""

    def __init__(self, table):
        self.t = table
        self.shape = table.shape

    def ptr_ld(self):
        k, ld = self.t.shape
        return C.c_void_p(self.t.data_ptr() + (k - 1) * ld * 4), -ld


def _nbr_ptr_ld(nbr):
    if nbr is None:
        return _ptr(None), 0
    if isinstance(nbr, RevNbr):
        return nbr.ptr_ld()
    return _ptr(nbr), nbr.shape[1]


def spconv_fwd(inp, w, nbr, n_out_dev, n_out, cout, transpose_w=False, tag=None, addend=None):
    """out[m] = sum_k in[nbr[k][m]] @ W[k]; w: [K, Cin_w, Cout_w] contiguous. Returns [n_out, cout].  nbr may be a RevNbr.
    addend (bf16 [n_out, cout], optional): summed into the result - by the kernel's epilogue where it has one, else afterwards."""
    kvol = w.shape[0]
    cin = inp.shape[1]
    out = torch.empty((n_out, cout), dtype=inp.dtype, device=inp.device)
    nbr_p, ld = _nbr_ptr_ld(nbr)
    if isinstance(nbr, RevNbr):
        nbr = nbr.t.flip(0) if (TIMER is not None and TIMER.mode == "census") else None      # only the census counts pairs from it
    t = TIMER
    e0 = t.begin() if t is not None else None
    rc = -2
    if addend is not None and inp.dtype == torch.bfloat16 and USE_IGEMM_V2 and ld != 0 and addend.dtype == torch.bfloat16 \
            and addend.shape == out.shape and addend.is_contiguous():
        rc = lib().u3d_igemm_fwd_add_bf16(_ptr(inp), _ptr(w), nbr_p, ld, _ptr(addend), _ptr(out), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                          1 if transpose_w else 0, _stream())
        if rc not in (0, -2):
            _check(rc, "igemm_fwd_add_bf16")
        if rc == 0:
            addend = None                      # already in
    if rc == -2 and inp.dtype == torch.bfloat16 and USE_IGEMM_V2:
        rc = lib().u3d_igemm_fwd_bf16(_ptr(inp), _ptr(w), nbr_p, ld, _ptr(out), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                      1 if transpose_w else 0, _stream())
        if rc not in (0, -2):
            _check(rc, "igemm_fwd_bf16")
    if rc == -2:      # shape served by the first-generation kernel (small channel counts, f32)
        _check(lib().u3d_spconv_fwd(_ptr(inp), _ptr(w), nbr_p, ld, _ptr(out), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                    1 if transpose_w else 0, dtype_code(inp), _stream()), "spconv_fwd")
    if t is not None:
        meta = None
        if t.mode == "census":
            pairs = int((nbr[:, :n_out] >= 0).sum().item()) if nbr is not None else n_out
            s = inp.element_size()
            meta = dict(kind=CALL_KIND, v2=bool(rc == 0), n_in=inp.shape[0], n_out=n_out, cin=cin, cout=cout, kvol=kvol, pairs=pairs,
                        bytes=inp.shape[0] * cin * s + n_out * cout * s + 8 * pairs + kvol * cin * cout * s,
                        flops=2 * pairs * cin * cout)
        t.end(tag or ("spconv_dgrad" if transpose_w else "spconv_fwd"), e0, meta)
    if addend is not None:
        out += addend
    return out


def split_rows(x, n_dev, n_cap=None):
    """f32 [n_cap, c] -> bf16 [2 * n_cap, c]: hi plane = bf16(x), lo plane = bf16(x - hi) (u3d_split_rows_f32).  Rows between the
    device-side count and n_cap are written as ZEROS in both planes (the kernel zero-fills up to the capacity: the LDS-DMA kernels stage
    whole tiles, and a NaN bit pattern in a never-named padding row would still poison a BatchNorm statistic through 0 * NaN)."""
    n_cap = x.shape[0] if n_cap is None else n_cap
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((2 * n_cap, x.shape[1]), dtype=torch.bfloat16, device=x.device)
    _check(lib().u3d_split_rows_f32(_ptr(x), _ptr(n_dev), n_cap, x.shape[1], _ptr(out), _stream()), "split_rows_f32")
    return out


def spconv_fwd_split_direct(xs, w3, nbr, n_out_dev, n_out, cout, tag="spconv_fwd"):
    """Split-bf16 product on a narrow 27-offset level (u3d_igemm_direct_split_bf16): xs bf16 planes [2 * n_in, cin], w3 bf16
    [81, cout, cin] = (wh, wl, wh), nbr the PLAIN table (or a RevNbr) -> f32 [n_out, cout]."""
    cin, n_in = xs.shape[1], xs.shape[0] // 2
    out = torch.empty((n_out, cout), dtype=torch.float32, device=xs.device)
    nbr_p, ld = _nbr_ptr_ld(nbr)
    t = TIMER
    e0 = t.begin() if t is not None else None
    _check(lib().u3d_igemm_direct_split_bf16(_ptr(xs), _ptr(w3), nbr_p, ld, _ptr(out), _ptr(n_out_dev), n_out, n_in, cin, cout, _stream()),
           "igemm_direct_split_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            tab = nbr.t.flip(0) if isinstance(nbr, RevNbr) else nbr
            pairs = int((tab[:, :n_out] >= 0).sum().item())
            meta = dict(kind=CALL_KIND, v2=True, split=True, n_in=n_in, n_out=n_out, cin=cin, cout=cout, kvol=27, pairs=pairs,
                        bytes=n_in * cin * 4 + n_out * cout * 4 + 8 * pairs + 27 * cin * cout * 4, flops=2 * pairs * cin * cout)
        t.end(tag, e0, meta)
    return out


class _SplitRowsJobs(C.Structure):
    _fields_ = [("src", C.c_void_p * 32), ("dst", C.c_void_p * 32), ("ld", C.c_int32 * 32), ("rows", C.c_int32 * 32), ("cols", C.c_int32 * 32),
                ("first_block", C.c_int32 * 33), ("njobs", C.c_int32)]


def split_rows_batch_into(tensors, outs):
    """hi / lo bf16 planes of several f32 row matrices (2-D, unit column stride, any row stride) in one launch per 32 of them
    (u3d_split_rows_batch); outs[i]: bf16 [2 * rows, cols] (hi plane, then lo plane), written here."""
    for i in range(0, len(tensors), 32):
        jobs = _SplitRowsJobs()
        chunk = list(zip(tensors[i:i + 32], outs[i:i + 32]))
        jobs.njobs = len(chunk)
        for j, (t, o) in enumerate(chunk):
            assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0
            assert o.dtype == torch.bfloat16 and o.is_contiguous() and tuple(o.shape) == (2 * t.shape[0], t.shape[1])
            jobs.src[j], jobs.dst[j] = t.data_ptr(), o.data_ptr()
            jobs.ld[j], jobs.rows[j], jobs.cols[j] = t.stride(0), t.shape[0], t.shape[1]
        _check(lib().u3d_split_rows_batch(C.byref(jobs), _stream()), "split_rows_batch")


def split_rows_batch(tensors):
    outs = [torch.empty((2 * t.shape[0], t.shape[1]), dtype=torch.bfloat16, device=t.device) for t in tensors]
    split_rows_batch_into(tensors, outs)
    return outs


def sum3(a, b, c):
    """(a + b) + c for three contiguous f32 tensors of one shape, one launch (u3d_sum3_f32)."""
    assert a.shape == b.shape == c.shape and a.dtype == b.dtype == c.dtype == torch.float32
    a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
    if a.numel() % 4 != 0 or not a.is_cuda:
        return a + b + c
    out = torch.empty_like(a)
    _check(lib().u3d_sum3_f32(_ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _stream()), "sum3_f32")
    return out


def _split3_geometry(w, layout, nmajor):
    """(k, a, b, sk, sa, sb): the [K, A, B] view of a conv parameter in its checkpoint layout, as element strides."""
    if layout == "dhwio":
        kd, kh, kw, cin, cout = w.shape
        k = kd * kh * kw
        sk, s_ci, s_co = cin * cout, cout, 1
    else:
        cout, cin, kd, kh, kw = w.shape
        k = kd * kh * kw
        sk, s_ci, s_co = 1, k, cin * k
    a, b, sa, sb = (cout, cin, s_co, s_ci) if nmajor else (cin, cout, s_ci, s_co)
    return k, a, b, sk, sa, sb


class Split3Set:
    """Every (conv parameter, layout, n-major?) triple a step asks for, refreshed by ONE launch at the top of the step
    (u3d_split3_weights_batch).  A request the set has not seen yet is served by its own launch and joins the set; entries whose
    parameter died or moved (TrainStep re-homes parameters into its flat buffer) are dropped / re-described at the next refresh."""

    class _Job(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("sk", C.c_int64), ("sa", C.c_int64), ("sb", C.c_int64),
                    ("K", C.c_int32), ("A", C.c_int32), ("B", C.c_int32), ("first_block", C.c_int32)]

    def __init__(self):
        self.entries = {}            # (data_ptr, nmajor) -> [weakref, layout, nmajor, dst, data_ptr, in the job table?, version, shape]
        self.jobs_dev = None
        self._keep = []              # every job table ever uploaded: a captured graph may still read an older one
        self.total_blocks = 0
        self.dirty = True
        self.fresh = False           # the dst tensors hold the split of the CURRENT parameter values

    def get(self, weight, layout, nmajor):
        """Keyed by the parameter's MEMORY (the backward sees the saved tensor, not necessarily the Parameter object)."""
        e = self.entries.get((weight.data_ptr(), bool(nmajor)))
        if e is not None and e[0]() is not None and e[1] == layout and e[7] == tuple(weight.shape):
            # served only when THIS step's refresh computed it from the parameter's current values (in-place torch updates bump the
            # version; TrainStep's flat AdamW does not, but every step of it starts with a refresh)
            if self.fresh and e[5] and e[6] == weight._version:
                return e[3]
        return None

    def add(self, weight, layout, nmajor, dst):
        import weakref
        key = (weight.data_ptr(), bool(nmajor))
        old = self.entries.get(key)
        if old is not None and old[0]() is not None and old[1] == layout and old[7] == tuple(weight.shape) and old[5]:
            return                    # already in the job table (asked for before this step's refresh could serve it)
        self.entries[key] = [weakref.ref(weight), layout, bool(nmajor), dst, weight.data_ptr(), False, -1, tuple(weight.shape)]
        self.dirty = True

    def refresh(self):
        """One launch over every live entry (called at the top of a step's forward, inside the captured graph too - the job table is
        rebuilt on the host only OUTSIDE a capture; a table that went stale during one falls back to per-request launches)."""
        dead = [k for k, e in self.entries.items() if e[0]() is None]
        for k in dead:
            del self.entries[k]
            self.dirty = True
        moved = [k for k, e in self.entries.items() if e[0]().data_ptr() != e[4]]      # re-homed parameters (TrainStep's flat buffer): asked for again
        for k in moved:
            del self.entries[k]
            self.dirty = True
        if not self.entries:
            self.fresh = False
            return
        if self.dirty:
            if torch.cuda.is_current_stream_capturing():
                self.fresh = False
                return
            assert int(lib().u3d_split3_job_bytes()) == C.sizeof(Split3Set._Job)
            jobs = (Split3Set._Job * len(self.entries))()
            fb = 0
            for j, e in enumerate(self.entries.values()):
                w = e[0]().detach()
                k, a, b, sk, sa, sb = _split3_geometry(w, e[1], e[2])
                jobs[j].src, jobs[j].dst = w.data_ptr(), e[3].data_ptr()
                jobs[j].sk, jobs[j].sa, jobs[j].sb = sk, sa, sb
                jobs[j].K, jobs[j].A, jobs[j].B, jobs[j].first_block = k, a, b, fb
                fb += int(lib().u3d_split3_job_blocks(k, a, b))
                e[5] = True
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
            dev = next(iter(self.entries.values()))[3].device
            self.jobs_dev = raw.to(dev)
            self._keep.append(self.jobs_dev)
            self.total_blocks, self.njobs, self.dirty = fb, len(self.entries), False
        _check(lib().u3d_split3_weights_batch(_ptr(self.jobs_dev), self.njobs, self.total_blocks, _stream()), "split3_weights_batch")
        for e in self.entries.values():
            e[6] = e[0]()._version
        self.fresh = True


SPLIT3_DEFAULT = Split3Set()   # scopes without an owner (ad-hoc calls, tests); a detector brings its own set (detector.stage_features)
SPLIT3_ACTIVE = None           # the set of the split scope being executed (sparse.split_scope), None outside
SPLIT3_BATCH = True            # test-only module attribute: False = one u3d_split3_weights launch per request (the A/B formulation)


def split3_weights(weight, layout, nmajor, cache=None):
    """Conv PARAMETER (f32, checkpoint layout "dhwio" [kD,kH,kW,Cin,Cout] or "oidhw" [Cout,Cin,kD,kH,kW]) -> bf16 [3K, A, B] = (hi, lo, hi)
    of its [K, Cout, Cin] (nmajor) or [K, Cin, Cout] view, read in place through element strides (u3d_split3_weights) - or, when the
    step's batched refresh (Split3Set) already produced it, the cached tensor."""
    cache = cache if cache is not None else SPLIT3_ACTIVE
    if not (SPLIT3_BATCH and weight.is_cuda):
        cache = None
    if cache is not None:
        hit = cache.get(weight, layout, nmajor)
        if hit is not None:
            return hit
    w = weight.detach()
    assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 5
    k, a, b, sk, sa, sb = _split3_geometry(w, layout, nmajor)
    out = torch.empty((3 * k, a, b), dtype=torch.bfloat16, device=w.device)
    _check(lib().u3d_split3_weights(_ptr(w), sk, sa, sb, k, a, b, _ptr(out), _stream()), "split3_weights")
    # only PARAMETERS join the set: a temporary (the FPN's transposed-conv weights reach here as a re-laid-out copy made each step) has
    # new memory every time - its entry would be pruned and re-added at every refresh and keep the job table dirty for ever
    if cache is not None and isinstance(weight, torch.nn.Parameter) and not torch.cuda.is_current_stream_capturing():
        cache.add(weight, layout, nmajor, out)
    return out


def spconv_fwd_split(xs, w3, nbr3, n_out_dev, n_out, cout, want_stats=False, tag="spconv_fwd", addend=None):
    """Split-bf16 product (u3d_igemm_fwd_split_bf16): xs bf16 [2 * n_in, cin] planes, w3 bf16 [3K, cout, cin] = (wh, wl, wh), nbr3 int32
    [3K, ld] = (nbr, nbr, nbr + n_in) -> f32 [n_out, cout] (+ per-tile BatchNorm sums f64 [tiles, 2, cout], rows per tile)."""
    kvol3, cin = w3.shape[0], xs.shape[1]
    out = torch.empty((n_out, cout), dtype=torch.float32, device=xs.device)
    stats, tr = None, 0
    if want_stats:
        tr = int(lib().u3d_igemm_fwd_stats_rows(n_out, cin, cout, kvol3))
        if tr:
            stats = torch.empty(((n_out + tr - 1) // tr, 2, cout), dtype=torch.float64, device=xs.device)
    t = TIMER
    e0 = t.begin() if t is not None else None
    assert addend is None or (addend.dtype == torch.float32 and tuple(addend.shape) == (n_out, cout))
    _check(lib().u3d_igemm_fwd_split_bf16(_ptr(xs), _ptr(w3), _ptr(nbr3), nbr3.shape[1], _ptr(out), _ptr(n_out_dev), n_out, cin, cout, kvol3,
                                          _ptr(stats), _ptr(addend), _stream()), "igemm_fwd_split_bf16")
    if t is not None:
        meta = None
        if t.mode == "census":
            k = kvol3 // 3
            pairs = int((nbr3[:k, :n_out] >= 0).sum().item())
            # priced as the f32 convolution it stands for: f32 rows in and out, f32 weights, 2 * pairs * cin * cout flops (the kernel issues 3x)
            meta = dict(kind=CALL_KIND, v2=True, split=True, n_in=xs.shape[0] // 2, n_out=n_out, cin=cin, cout=cout, kvol=k, pairs=pairs,
                        bytes=(xs.shape[0] // 2) * cin * 4 + n_out * cout * 4 + 8 * pairs + k * cin * cout * 4, flops=2 * pairs * cin * cout)
        t.end(tag, e0, meta)
    return (out, stats, tr) if want_stats else out


def spconv_wgrad(inp, dout, nbr, n_out_dev, kvol, out_oik=False, out=None):
    """-> f32 [K, Cin, Cout]; with out_oik (bf16 second-generation path only): [Cout, Cin, K] (nn.Conv3d's layout).
    out: contiguous f32 tensor of kvol*cin*cout elements to write into (bf16 path) - returned viewed in the result's shape."""
    cin, cout, n_out = inp.shape[1], dout.shape[1], dout.shape[0]
    v2 = bool(inp.dtype == torch.bfloat16 and USE_IGEMM_V2 and cin % 16 == 0 and cout % 16 == 0)
    assert v2 or not out_oik, "out_oik needs the bf16 implicit-GEMM weight-gradient path"
    shape = (cout, cin, kvol) if out_oik else (kvol, cin, cout)
    if out is not None and v2 and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == kvol * cin * cout:
        dw = out.view(shape)
    else:
        dw = torch.empty(shape, dtype=torch.float32, device=inp.device)
    t = TIMER
    meta = None
    if t is not None and t.mode == "census":
        pairs = int((nbr[:, :n_out] >= 0).sum().item()) if nbr is not None else n_out
        s = inp.element_size()
        meta = dict(kind=CALL_KIND, v2=bool(inp.dtype == torch.bfloat16 and USE_IGEMM_V2 and cin % 16 == 0 and cout % 16 == 0),
                    n_in=inp.shape[0], n_out=n_out, cin=cin, cout=cout, kvol=kvol, pairs=pairs,
                    bytes=inp.shape[0] * cin * s + n_out * cout * s + 8 * pairs + kvol * cin * cout * 4, flops=2 * pairs * cin * cout)
    if inp.dtype == torch.bfloat16 and USE_IGEMM_V2 and ((cin % 16 == 0 and cout % 16 == 0) or (cin == 8 and cout == 16 and not out_oik and kvol <= 27)):
        wsb = int(lib().u3d_igemm_wgrad_bf16_workspace(n_out, cin, cout, kvol))
        ws = torch.empty(wsb, dtype=torch.uint8, device=inp.device)
        ld = nbr.shape[1] if nbr is not None else 0
        e0 = t.begin() if t is not None else None
        rc = lib().u3d_igemm_wgrad_bf16(_ptr(inp), _ptr(dout), _ptr(nbr), ld, _ptr(dw), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                        1 if out_oik else 0, _ptr(ws), wsb, _stream())
        if rc != -2:
            _check(rc, "igemm_wgrad_bf16")
            if t is not None:
                t.end("spconv_wgrad", e0, meta)
            return dw
        # a channel pair the implicit-GEMM weight-gradient kernels do not serve (e.g. 32 -> 16): the first-generation kernel below
        gen = spconv_wgrad_generic(inp, dout, nbr, n_out_dev, kvol)
        dw.copy_(gen.permute(2, 1, 0) if out_oik else gen)
        if t is not None:
            t.end("spconv_wgrad", e0, meta)
        return dw
    wsb = int(lib().u3d_spconv_wgrad_workspace(n_out, cin, cout, kvol))
    ws = torch.empty(wsb, dtype=torch.uint8, device=inp.device)
    ld = nbr.shape[1] if nbr is not None else 0
    e0 = t.begin() if t is not None else None
    _check(lib().u3d_spconv_wgrad(_ptr(inp), _ptr(dout), _ptr(nbr), ld, _ptr(dw), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                  dtype_code(inp), _ptr(ws), wsb, _stream()), "spconv_wgrad")
    if t is not None:
        t.end("spconv_wgrad", e0, meta)
    return dw


def spconv_wgrad_generic(inp, dout, nbr, n_out_dev, kvol):
    """First-generation weight gradient (any channel counts, f32 or bf16): f32 [K, Cin, Cout]."""
    cin, cout, n_out = inp.shape[1], dout.shape[1], dout.shape[0]
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=inp.device)
    wsb = int(lib().u3d_spconv_wgrad_workspace(n_out, cin, cout, kvol))
    ws = torch.empty(wsb, dtype=torch.uint8, device=inp.device)
    ld = nbr.shape[1] if nbr is not None else 0
    _check(lib().u3d_spconv_wgrad(_ptr(inp), _ptr(dout), _ptr(nbr), ld, _ptr(dw), _ptr(n_out_dev), n_out, cin, cout, kvol,
                                  dtype_code(inp), _ptr(ws), wsb, _stream()), "spconv_wgrad")
    return dw


def colsum(x):
    """f32 column sums of a dense [n, C] f32/bf16 matrix (fixed order; safe under HIP-graph replay, unlike torch's sum(0))."""
    x = x if x.is_contiguous() else x.contiguous()
    n, c = x.shape
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    wsb = int(lib().u3d_colsum_workspace(n, c))
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=x.device)
    _check(lib().u3d_colsum(_ptr(x), n, c, dtype_code(x), _ptr(out), _ptr(ws), wsb, _stream()), "colsum")
    return out


def bn_stats(x, n_dev):
    n, c = x.shape
    sums = torch.empty((2, c), dtype=torch.float64, device=x.device)
    wsb = int(lib().u3d_bn_stats_workspace(n, c))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _check(lib().u3d_bn_stats(_ptr(x), _ptr(n_dev), n, c, dtype_code(x), _ptr(sums), _ptr(ws), wsb, _stream()), "bn_stats")
    return sums


def bn_finalize(sums, n_dev, n_cap, eps, momentum, running_mean=None, running_var=None, num_batches=None):
    c = sums.shape[1]
    mean = torch.empty((c,), dtype=torch.float32, device=sums.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=sums.device)
    _check(lib().u3d_bn_finalize(_ptr(sums), _ptr(n_dev), n_cap, c, eps, momentum, _ptr(running_mean), _ptr(running_var),
                                 _ptr(num_batches), _ptr(mean), _ptr(invstd), _stream()), "bn_finalize")
    return mean, invstd


def bn_forward_stats(x, n_dev, eps, momentum, running_mean=None, running_var=None, num_batches=None):
    """Training statistics of a row matrix: (mean, invstd) f32 [C]; updates the running statistics like nn.BatchNorm1d."""
    n, c = x.shape
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    wsb = int(lib().u3d_bn_stats_workspace(n, c))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
    _check(lib().u3d_bn_forward_stats(_ptr(x), _ptr(n_dev), n, c, dtype_code(x), eps, momentum, _ptr(running_mean), _ptr(running_var),
                                      _ptr(num_batches), _ptr(mean), _ptr(invstd), _ptr(ws), ws.numel(), _stream()), "bn_forward_stats")
    return mean, invstd


def bn_apply(x, mean, invstd, gamma, beta, residual, relu, n_dev, row_map=None, post_add=None, want_planes=False):
    """row_map (int32 [n], a bijection): row r of x lands in row row_map[r] of y (see include/u3d_hip.h).
    want_planes (f32 rows): -> (y, planes) with planes bf16 [2 * n, c] = split_rows(y) written by the same pass (u3d_bn_apply_planes), or
    (y, None) for a shape the fused pass does not take."""
    n, c = x.shape
    y = torch.empty_like(x)
    if want_planes:
        planes = None
        if x.dtype == torch.float32 and x.is_cuda:
            planes = torch.empty((2 * n, c), dtype=torch.bfloat16, device=x.device)
            rc = lib().u3d_bn_apply_planes(_ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(residual), int(relu), _ptr(y),
                                           _ptr(planes), _ptr(n_dev), n, c, _ptr(row_map), _ptr(post_add), _stream())
            if rc == 0:
                return y, planes
            if rc != -2:
                _check(rc, "bn_apply_planes")
            planes = None
        _check(lib().u3d_bn_apply(_ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(residual), int(relu),
                                  _ptr(y), _ptr(n_dev), n, c, dtype_code(x), _ptr(row_map), _ptr(post_add), _stream()), "bn_apply")
        return y, None
    _check(lib().u3d_bn_apply(_ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(residual), int(relu),
                              _ptr(y), _ptr(n_dev), n, c, dtype_code(x), _ptr(row_map), _ptr(post_add), _stream()), "bn_apply")
    return y


def bn_bwd_stats(dy, y, x, mean, invstd, relu, n_dev, gamma=None, beta=None, row_map=None, want_f32=False):
    """y may be None (relu, no residual in the forward): the ReLU mask is recomputed from x with gamma/beta."""
    n, c = x.shape
    sums = torch.empty((2, c), dtype=torch.float64, device=x.device)
    wsb = int(lib().u3d_bn_stats_workspace(n, c))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    s32 = torch.empty((2, c), dtype=torch.float32, device=x.device) if want_f32 else None
    _check(lib().u3d_bn_bwd_stats(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), int(relu), _ptr(n_dev), n, c,
                                  dtype_code(x), _ptr(sums), _ptr(ws), wsb, _ptr(row_map), _ptr(s32), _stream()), "bn_bwd_stats")
    return (sums, s32) if want_f32 else sums


def bn_bwd_apply(dy, y, x, mean, invstd, gamma, sums, relu, n_dev, want_dres, beta=None, row_map=None, want_planes=False):
    """want_planes (f32 rows): -> (dx, dres, planes of dx or None) - u3d_bn_bwd_apply_planes."""
    n, c = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    if want_planes:
        if x.dtype == torch.float32 and x.is_cuda:
            planes = torch.empty((2 * n, c), dtype=torch.bfloat16, device=x.device)
            rc = lib().u3d_bn_bwd_apply_planes(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(sums), int(relu),
                                               _ptr(dx), _ptr(dres), _ptr(planes), _ptr(n_dev), n, c, _ptr(row_map), _stream())
            if rc == 0:
                return dx, dres, planes
            if rc != -2:
                _check(rc, "bn_bwd_apply_planes")
        _check(lib().u3d_bn_bwd_apply(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(sums), int(relu),
                                      _ptr(dx), _ptr(dres), _ptr(n_dev), n, c, dtype_code(x), _ptr(row_map), _stream()), "bn_bwd_apply")
        return dx, dres, None
    _check(lib().u3d_bn_bwd_apply(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(sums), int(relu),
                                  _ptr(dx), _ptr(dres), _ptr(n_dev), n, c, dtype_code(x), _ptr(row_map), _stream()), "bn_bwd_apply")
    return dx, dres


def to_dense(feat, coors, n_dev, batch, dims):
    """-> logical [B, C, D, H, W] tensor in channels_last_3d memory format."""
    n, c = feat.shape
    dz, dy, dx = dims
    vol = torch.zeros((batch, dz, dy, dx, c), dtype=feat.dtype, device=feat.device)
    _check(lib().u3d_to_dense(_ptr(feat), _ptr(coors), _ptr(n_dev), n, c, _ptr(vol), dz, dy, dx, dtype_code(feat), _stream()),
           "to_dense")
    return vol.permute(0, 4, 1, 2, 3)


def from_dense(vol_cl, coors, n_dev, n):
    """vol_cl: contiguous [B, Dz, Dy, Dx, C] -> rows [n, C]."""
    b, dz, dy, dx, c = vol_cl.shape
    feat = torch.empty((n, c), dtype=vol_cl.dtype, device=vol_cl.device)
    _check(lib().u3d_from_dense(_ptr(vol_cl), _ptr(coors), _ptr(n_dev), n, c, _ptr(feat), dz, dy, dx, dtype_code(vol_cl),
                                _stream()), "from_dense")
    return feat


def gather_rows(inp, idx):
    n = idx.shape[0]
    out = torch.empty((n,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    row_bytes = inp[0].numel() * inp.element_size()
    _check(lib().u3d_gather_rows(_ptr(inp), _ptr(idx), n, row_bytes, _ptr(out), _stream()), "gather_rows")
    return out


def scatter_rows(inp, idx, n_out):
    out = torch.zeros((n_out,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    row_bytes = inp[0].numel() * inp.element_size()
    _check(lib().u3d_scatter_rows(_ptr(inp), _ptr(idx), idx.shape[0], row_bytes, _ptr(out), _stream()), "scatter_rows")
    return out


# --------------------------------------------------------------------------------------------------
# query side: FPS, matching, IoU
# --------------------------------------------------------------------------------------------------
FPS_REG_MAX = 20480


def fps_err_buffer(device):
    """int32 [2] time-out record of FPS calls over sets above FPS_REG_MAX points (include/u3d_hip.h u3d_fps): [0] = this call timed
    out (its samples must not be used), [1] = calls that timed out so far.  Owned by the caller (TrainStep keeps one and feeds [0]
    into the step's collective hold flag); reading it synchronises with the device."""
    return torch.zeros(2, dtype=torch.int32, device=device)


def _fps_err_arg(err, max_n, device):
    """The several-workgroup form REQUIRES the record; smaller sets never write it (and a captured step saves the clearing launch)."""
    if max_n <= FPS_REG_MAX:
        return None
    return fps_err_buffer(device) if err is None else err


def fps(base, set_off, set_n, max_n, m, err=None, poll_ticks=0, max_wg=0):
    """base: f32 device buffer; set_off int64 [S] element offsets; set_n int32 [S]; -> idx int32 [S, m].
    err: int32 [2] from fps_err_buffer() (None: a private one, reachable as `idx._u3d_fps_err`); poll_ticks / max_wg: the C ABI's
    time-out limit (100 MHz ticks, 0 = 0.5 s) and resident-workgroup budget (0 = 3/4 of the device's CUs, 1 = streaming kernel)."""
    nsets = set_off.shape[0]
    out = torch.empty((nsets, m), dtype=torch.int32, device=base.device)
    temp, stride = None, 0
    if max_n > FPS_REG_MAX:
        temp = torch.empty((nsets, max_n), dtype=torch.float32, device=base.device)
        stride = max_n
    err = _fps_err_arg(err, max_n, base.device)
    _check(lib().u3d_fps(_ptr(base), _ptr(set_off), _ptr(set_n), nsets, max_n, m, _ptr(out), _ptr(temp), stride, _ptr(err), int(poll_ticks),
                         int(max_wg), _stream()), "fps")
    out._u3d_fps_err = err
    return out


def fps_queries(points, coors, scene_off, voxel_off, batch, max_n, m, err=None, poll_ticks=0, max_wg=0):
    """The detector's two FPS passes + their glue in three launches (u3d_fps_prep | u3d_fps2 | u3d_fps_points).
    points f32 [N,F] (packed-triple view: the reference hands the whole [N,F] buffer to the sampler), coors int32 [V,4] (b,z,y,x),
    scene_off / voxel_off int32 [B+1] -> fpsbpts f32 [B, 2m, 3] in the unit cube (ref: uni3detr.py:178-189).  err / poll_ticks /
    max_wg: as fps()."""
    dev = points.device
    F_ = points.shape[1]
    V = coors.shape[0]
    vox = torch.empty((max(V, 1), 3), dtype=torch.float32, device=dev)
    set_off = torch.empty((2 * batch,), dtype=torch.int64, device=dev)
    set_n = torch.empty((2 * batch,), dtype=torch.int32, device=dev)
    _check(lib().u3d_fps_prep(_ptr(coors), V, _ptr(scene_off), _ptr(voxel_off), batch, F_, _ptr(vox), _ptr(set_off), _ptr(set_n), _stream()),
           "fps_prep")
    idx = torch.empty((2 * batch, m), dtype=torch.int32, device=dev)
    temp, stride = None, 0
    if max_n > FPS_REG_MAX:
        temp = torch.empty((2 * batch, max_n), dtype=torch.float32, device=dev)
        stride = max_n
    err = _fps_err_arg(err, max_n, dev)
    _check(lib().u3d_fps2(_ptr(points), _ptr(vox), batch, _ptr(set_off), _ptr(set_n), 2 * batch, max_n, m, _ptr(idx), _ptr(temp), stride,
                          _ptr(err), int(poll_ticks), int(max_wg), _stream()), "fps2")
    idx._u3d_fps_err = err
    out = torch.empty((batch, 2 * m, 3), dtype=torch.float32, device=dev)
    _check(lib().u3d_fps_points(_ptr(points), F_, _ptr(vox), _ptr(idx), _ptr(scene_off), _ptr(voxel_off), batch, m, _ptr(out), _stream()),
           "fps_points")
    return out, idx


def loss_targets(asg, gt, labels, gt_off, ncls):
    """asg int32 [L,B,Q], gt f32 [sumG,gd], labels int32 [sumG] -> (asg int64, w f32 [L,B,Q], tgt f32 [L,B,Q,gd], lab int64, num_pos f32 [L])."""
    L, B, Q = asg.shape
    gd = gt.shape[1]
    dev = asg.device
    a64 = torch.empty((L, B, Q), dtype=torch.int64, device=dev)
    w = torch.empty((L, B, Q), dtype=torch.float32, device=dev)
    tgt = torch.empty((L, B, Q, gd), dtype=torch.float32, device=dev)
    lab = torch.empty((L, B, Q), dtype=torch.int64, device=dev)
    npos = torch.empty((L,), dtype=torch.float32, device=dev)
    _check(lib().u3d_loss_targets(_ptr(asg), _ptr(gt), _ptr(labels), _ptr(gt_off), L, B, Q, gd, ncls, _ptr(a64), _ptr(w), _ptr(tgt), _ptr(lab),
                                  _ptr(npos), _stream()), "loss_targets")
    return a64, w, tgt, lab, npos


def query_embed_fwd(tgt, anchor, fps, rnd, groups):
    """-> (query_embeds [B,G*nq,c+3], query [B,G*nq,c], ref_logits [B,G*nq,3], sigmoid(ref_logits)) f32 (u3d_query_embed_fwd)."""
    B, nq, c = fps.shape[0], anchor.shape[0], tgt.shape[1]
    dev = tgt.device
    qe = torch.empty((B, groups * nq, c + 3), dtype=torch.float32, device=dev)
    q = torch.empty((B, groups * nq, c), dtype=torch.float32, device=dev)
    r = torch.empty((B, groups * nq, 3), dtype=torch.float32, device=dev)
    rs = torch.empty((B, groups * nq, 3), dtype=torch.float32, device=dev)
    _check(lib().u3d_query_embed_fwd(_ptr(tgt), _ptr(anchor), _ptr(fps), _ptr(rnd), B, nq, groups, c, _ptr(qe), _ptr(q), _ptr(r), _ptr(rs),
                                     _stream()), "query_embed_fwd")
    return qe, q, r, rs


def query_embed_bwd(dqe, dq, dr, drs, anchor, B, nq, groups, c, device):
    dt = torch.empty((2 * nq, c), dtype=torch.float32, device=device)
    da = torch.empty((nq, 3), dtype=torch.float32, device=device)
    _check(lib().u3d_query_embed_bwd(_ptr(dqe), _ptr(dq), _ptr(dr), _ptr(drs), _ptr(anchor), B, nq, groups, c, _ptr(dt), _ptr(da), _stream()),
           "query_embed_bwd")
    return dt, da


def match_cost(cls, box, gt, labels, gt_off, gmax, w_cls, w_reg, w_iou, alpha=0.25, gamma=2.0):
    """cls [L,B,Q,C], box [L,B,Q,code] f32; gt [sumG,7] gravity-centre; labels int32; -> cost f32 [L*B, gmax, Q]."""
    L, B, Q, Ccls = cls.shape
    cost = torch.zeros((L * B, gmax, Q), dtype=torch.float32, device=cls.device)
    _check(lib().u3d_match_cost(_ptr(cls), _ptr(box), _ptr(gt), _ptr(labels), _ptr(gt_off), L, B, Q, Ccls, box.shape[-1], gmax,
                                w_cls, w_reg, w_iou, alpha, gamma, _ptr(cost), _stream()), "match_cost")
    return cost


def lsa(cost, gt_off, L, B, Q, nq, gmax):
    assigned = torch.empty((L, B, Q), dtype=torch.int32, device=cost.device)
    _check(lib().u3d_lsa(_ptr(cost), _ptr(gt_off), L, B, Q, nq, gmax, _ptr(assigned), _stream()), "lsa")
    return assigned


def iou3d_rotated_aligned(a, b):
    n = a.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=a.device)
    _check(lib().u3d_iou3d_rotated_aligned(_ptr(a), _ptr(b), n, _ptr(out), _stream()), "iou3d_rotated_aligned")
    return out


def trilinear_fwd(value_rows, grid, batch, dims):
    """value_rows [B*D*H*W, C] contiguous; grid f32 [B,N,3] -> [B,N,C]."""
    nq, c = grid.shape[1], value_rows.shape[1]
    out = torch.empty((batch, nq, c), dtype=value_rows.dtype, device=value_rows.device)
    _check(lib().u3d_trilinear_fwd(_ptr(value_rows), _ptr(grid), batch, nq, dims[0], dims[1], dims[2], c, _ptr(out),
                                   dtype_code(value_rows), _stream()), "trilinear_fwd")
    return out


def trilinear_bwd(value_rows, grid, dout, batch, dims, want_dvalue=True, want_dgrid=True, dvalue_accum=None):
    """dvalue_accum: an existing f32 [rows, C] buffer to ACCUMULATE the value gradient into (the kernel adds with atomics)."""
    nq, c = grid.shape[1], value_rows.shape[1]
    if want_dvalue:
        dvalue = dvalue_accum if dvalue_accum is not None else torch.zeros(value_rows.shape, dtype=torch.float32, device=value_rows.device)
    else:
        dvalue = None
    dgrid = torch.empty((batch, nq, 3), dtype=torch.float32, device=value_rows.device) if want_dgrid else None
    _check(lib().u3d_trilinear_bwd(_ptr(value_rows), _ptr(grid), _ptr(dout), batch, nq, dims[0], dims[1], dims[2], c, _ptr(dvalue),
                                   _ptr(dgrid), dtype_code(value_rows), _stream()), "trilinear_bwd")
    return dvalue, dgrid


def nms3d_classwise(boxes, scores, labels, thr):
    """boxes [n,7], scores [n], labels [n] -> indices kept, ordered by (label asc, score desc) like the reference's
    per-class loop over mmcv.ops.nms3d."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order, :7].float().contiguous()
    l = labels[order].int().contiguous()
    keep = torch.empty((n,), dtype=torch.uint8, device=boxes.device)
    wsb = int(lib().u3d_nms3d_workspace(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=boxes.device)
    _check(lib().u3d_nms3d(_ptr(b), _ptr(l), n, float(thr), _ptr(keep), _ptr(ws), wsb, _stream()), "nms3d")
    kept = order[keep.bool()]
    return kept[torch.argsort(labels[kept], stable=True)]


_COUNT_CACHE = {}


def soft_nms_classwise(boxes, scores, labels, num_classes, sigma, prune):
    """-> (idx long [k] into the input, decayed scores [k], labels long [k]) class-major, selection order inside a class."""
    n = boxes.shape[0]
    dev = boxes.device
    if n == 0:
        return (torch.zeros(0, dtype=torch.long, device=dev), torch.zeros(0, device=dev), torch.zeros(0, dtype=torch.long, device=dev))
    b = boxes[:, :7].contiguous().float()
    oi = torch.empty((num_classes, n), dtype=torch.int32, device=dev)
    os_ = torch.empty((num_classes, n), dtype=torch.float32, device=dev)
    oc = torch.empty((num_classes,), dtype=torch.int32, device=dev)
    _check(lib().u3d_soft_nms(_ptr(b), _ptr(scores.contiguous().float()), _ptr(labels.contiguous().int()), n, num_classes, float(sigma),
                              float(prune), _ptr(oi), _ptr(os_), _ptr(oc), _stream()), "soft_nms")
    sel = torch.arange(n, device=dev)[None, :] < oc[:, None]              # variable-size result: the one sync of this routine
    cls = torch.arange(num_classes, device=dev)[:, None].expand(-1, n)
    return oi[sel].long(), os_[sel], cls[sel]


def box_merge(boxes_sorted, labels_sorted, thr):
    """boxes f32 [n,7] sorted by descending score -> (merged [n,7], keep bool [n])."""
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    merged = torch.empty((n, 7), dtype=torch.float32, device=dev)
    keep = torch.zeros((n,), dtype=torch.uint8, device=dev)
    if n == 0:
        return merged, keep.bool()
    ws = torch.empty(max(int(lib().u3d_box_merge_workspace(n)), 16), dtype=torch.uint8, device=dev)
    _check(lib().u3d_box_merge(_ptr(boxes_sorted.contiguous().float()), _ptr(labels_sorted.contiguous().int()), n, float(thr), _ptr(merged),
                               _ptr(keep), _ptr(ws), ws.numel(), _stream()), "box_merge")
    return merged, keep.bool()


def count_tensor(n, device):
    """cached device int32 scalar holding n (row counts of static-size matrices)."""
    key = (str(device), int(n))
    t = _COUNT_CACHE.get(key)
    if t is None:
        t = torch.tensor([int(n)], dtype=torch.int32, device=device)
        _COUNT_CACHE[key] = t
    return t


def linear_bf16(x, w, bias, relu):
    """x bf16 [M,K], w bf16 [N,K] (nn.Linear layout), bias f32 [N]|None -> bf16 [M,N]; None if the shape is unsupported."""
    m, k = x.shape
    n = w.shape[0]
    if k % 64 or n % 64:
        return None
    out = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    _check(lib().u3d_linear_bf16(_ptr(x), _ptr(w), _ptr(bias), int(relu), _ptr(out), _ptr(count_tensor(m, x.device)), m, k, n, _stream()),
           "linear_bf16")
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0, workspace=None):
    """In-place clip + AdamW on flat f32 buffers; `state` = 16 zero-initialised device floats (see include/u3d_hip.h)."""
    n = param.numel()
    wsb = int(lib().u3d_adamw_workspace(n))
    ws = workspace if workspace is not None else torch.empty(wsb, dtype=torch.uint8, device=param.device)
    _check(lib().u3d_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, lr, betas[0], betas[1], eps, weight_decay,
                                float(max_norm), _ptr(state), _ptr(ws), ws.numel(), _stream()), "adamw_step")


def adamw_set_hyper(state, lr, betas, eps, weight_decay, max_norm):
    """hyper-parameters into slots [5..10] of the 16-float device state vector (stream-ordered, one tiny launch)."""
    _check(lib().u3d_adamw_set_hyper(_ptr(state), lr, betas[0], betas[1], eps, weight_decay, float(max_norm), _stream()), "adamw_set_hyper")


def adamw_step_state(param, grad, exp_avg, exp_avg_sq, state, skip=None, workspace=None, hold=None):
    """clip + AdamW with the hyper-parameters taken from `state` (see adamw_set_hyper); skip: uint8 per 64-element chunk or None;
    hold: one device float or None - > 0 turns the step into a no-op and counts it in state[12] (capacity overflow somewhere)."""
    n = param.numel()
    wsb = int(lib().u3d_adamw_workspace(n))
    ws = workspace if workspace is not None else torch.empty(wsb, dtype=torch.uint8, device=param.device)
    _check(lib().u3d_adamw_step_hold(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, _ptr(state), _ptr(skip), _ptr(hold),
                                     _ptr(ws), ws.numel(), _stream()), "adamw_step_hold")


def capacity_flag(counts, caps, flag):
    """flag[0] <- number of levels with counts[i][0] > caps[i] (counts: list of device int32 tensors, caps: python ints)."""
    n = len(counts)
    assert n == len(caps) and n <= 8 and flag.dtype == torch.float32
    _check(lib().u3d_capacity_flag(_ptr_array(counts), (C.c_int32 * max(n, 1))(*[int(c) for c in caps]), n, _ptr(flag), _stream()),
           "capacity_flag")


def tap_gather_sum(p, nbr, n_dev, n, c, kvol, addend=None):
    """din[i] = sum_k p[nbr[k][i], k*c:(k+1)*c] (+ addend[i]) (f32 accumulate); p: [n_out, kvol*c]."""
    out = torch.empty((n, c), dtype=p.dtype, device=p.device)
    if addend is not None:
        assert addend.shape == out.shape and addend.dtype == out.dtype and addend.is_contiguous()
    _check(lib().u3d_tap_gather_sum_add(_ptr(p), _ptr(nbr), nbr.shape[1], _ptr(n_dev), n, c, kvol, dtype_code(p), _ptr(addend), _ptr(out),
                                        _stream()), "tap_gather_sum")
    return out


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _batch_chunks(outs, maxb):
    """(start, end) slices of at most maxb batch slots that never split a run of slots naming the same output tensor."""
    o, n = 0, len(outs)
    while o < n:
        e = min(n, o + maxb)
        while e < n and e > o + 1 and outs[e].data_ptr() == outs[e - 1].data_ptr():
            e -= 1
        if e < n and outs[e].data_ptr() == outs[e - 1].data_ptr():
            raise U3DError("a group of batch slots with one output is larger than the batch limit")
        yield o, e
        o = e


def wgrad_batched(ins, douts, dws):
    """dws[b] <- ins[b]^T @ douts[b] for same-shape bf16 [M,Cin] / [M,Cout] pairs (f32 [Cin,Cout] outputs, preallocated).
    CONSECUTIVE slots that name the same output are summed into it (a weight used by several layers)."""
    m, cin = ins[0].shape
    cout = douts[0].shape[1]
    dev = ins[0].device
    MAXB = 48
    for o, e in _batch_chunks(dws, MAXB):
        a, b, c = ins[o:e], douts[o:e], dws[o:e]
        wsb = int(lib().u3d_wgrad_batched_workspace(len(a), m, cin, cout))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        _check(lib().u3d_wgrad_batched_bf16(_ptr_array(a), _ptr_array(b), _ptr_array(c), len(a), _ptr(count_tensor(m, dev)), m, cin, cout,
                                            _ptr(ws), ws.numel(), _stream()), "wgrad_batched_bf16")


def colsum_batched(xs, outs):
    """outs[b] <- column sums of xs[b] (same-shape [n, C] matrices, f32 [C] outputs, preallocated).  CONSECUTIVE slots that name the
    same output are summed into it."""
    n, c = xs[0].shape
    dev = xs[0].device
    MAXB = 64
    for o, e in _batch_chunks(outs, MAXB):
        a, b = xs[o:e], outs[o:e]
        wsb = int(lib().u3d_colsum_batched_workspace(len(a), n, c))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        _check(lib().u3d_colsum_batched(_ptr_array(a), _ptr_array(b), len(a), n, c, dtype_code(a[0]), _ptr(ws), ws.numel(), _stream()),
               "colsum_batched")


def layernorm_fwd(x2, gamma, beta, eps, relu, out_dtype):
    """x2 [n, C] f32/bf16 -> (y [n, C] out_dtype, mean [n], rstd [n])."""
    n, c = x2.shape
    y = torch.empty((n, c), dtype=out_dtype, device=x2.device)
    mean = torch.empty((n,), dtype=torch.float32, device=x2.device)
    rstd = torch.empty((n,), dtype=torch.float32, device=x2.device)
    _check(lib().u3d_layernorm_fwd(_ptr(x2), dtype_code(x2), n, c, _ptr(gamma), _ptr(beta), eps, int(relu), _ptr(y), dtype_code(y),
                                   _ptr(mean), _ptr(rstd), _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2, x2, gamma, beta, mean, rstd, relu):
    """-> (dx [n, C] in x2's dtype, partial f32 [2, nblocks, C] whose column sums are dgamma, dbeta)."""
    n, c = x2.shape
    dx = torch.empty_like(x2)
    nb = int(lib().u3d_layernorm_blocks(n))
    partial = torch.empty((2, nb, c), dtype=torch.float32, device=x2.device)
    _check(lib().u3d_layernorm_bwd(_ptr(dy2), dtype_code(dy2), _ptr(x2), dtype_code(x2), n, c, _ptr(gamma), _ptr(beta), _ptr(mean),
                                   _ptr(rstd), int(relu), _ptr(dx), _ptr(partial), _stream()), "layernorm_bwd")
    return dx, partial


def denormalize_boxes(codes):
    """codes [n, code>=8] f32 -> boxes [n, 7]."""
    codes = codes.contiguous()
    n, code = codes.shape
    out = torch.empty((n, 7), dtype=torch.float32, device=codes.device)
    _check(lib().u3d_denormalize_boxes(_ptr(codes), n, code, _ptr(out), _stream()), "denormalize_boxes")
    return out


def det_loss_fwd(cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, alpha, w_cls, w_box, w_iou, eps):
    L, m, c = cls.shape
    code, tdim = box.shape[-1], tgt.shape[-1]
    out = torch.empty((L, 4), dtype=torch.float32, device=cls.device)
    wsb = int(lib().u3d_det_loss_workspace(L, m))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=cls.device)
    _check(lib().u3d_det_loss_fwd(_ptr(cls), _ptr(box), _ptr(iou_logit), _ptr(tgt), _ptr(lab), _ptr(w), _ptr(iou_true), _ptr(cls_avg),
                                  _ptr(npos), _ptr(code_w), L, m, c, code, tdim, alpha, w_cls, w_box, w_iou, eps, _ptr(out), _ptr(ws),
                                  ws.numel(), _stream()), "det_loss_fwd")
    return out


def det_loss_bwd(cls, box, iou_logit, tgt, lab, w, iou_true, cls_avg, npos, code_w, gout, alpha, w_cls, w_box, w_iou, eps):
    L, m, c = cls.shape
    code, tdim = box.shape[-1], tgt.shape[-1]
    dcls, dbox, diou = torch.empty_like(cls), torch.empty_like(box), torch.empty_like(iou_logit)
    _check(lib().u3d_det_loss_bwd(_ptr(cls), _ptr(box), _ptr(iou_logit), _ptr(tgt), _ptr(lab), _ptr(w), _ptr(iou_true), _ptr(cls_avg),
                                  _ptr(npos), _ptr(code_w), _ptr(gout), L, m, c, code, tdim, alpha, w_cls, w_box, w_iou, eps, _ptr(dcls),
                                  _ptr(dbox), _ptr(diou), _stream()), "det_loss_bwd")
    return dcls, dbox, diou


def skinny_wgrad_partial(dy2, x2):
    """bf16 dy2 [m, n], x2 [m, k] with min(n, k) <= 16 -> f32 [chunks, n*k] whose column sums are dW [n, k]."""
    m, n = dy2.shape
    k = x2.shape[1]
    chunks = int(lib().u3d_skinny_wgrad_chunks(m))
    partial = torch.empty((chunks, n * k), dtype=torch.float32, device=dy2.device)
    _check(lib().u3d_skinny_wgrad_bf16(_ptr(dy2), _ptr(x2), m, n, k, _ptr(partial), _stream()), "skinny_wgrad_bf16")
    return partial


def skinny_wgrad_partial_batched(dys, xs):
    """The same for a list of (dy2 [m, n_i], x2 [m, k_i]) pairs over the same m rows: one launch per skinny side (<= 32 products each).
    -> list of f32 [chunks, n_i*k_i] partials, in input order."""
    m = dys[0].shape[0]
    chunks = int(lib().u3d_skinny_wgrad_chunks(m))
    outs = [torch.empty((chunks, d.shape[1] * x.shape[1]), dtype=torch.float32, device=d.device) for d, x in zip(dys, xs)]
    for dy_skinny in (1, 0):
        sel = [i for i, (d, x) in enumerate(zip(dys, xs)) if (d.shape[1] <= x.shape[1]) == bool(dy_skinny)]
        for o in range(0, len(sel), 32):
            ii = sel[o:o + 32]
            n = (C.c_int32 * len(ii))(*[dys[i].shape[1] for i in ii])
            k = (C.c_int32 * len(ii))(*[xs[i].shape[1] for i in ii])
            _check(lib().u3d_skinny_wgrad_batched(_ptr_array([dys[i] for i in ii]), _ptr_array([xs[i] for i in ii]),
                                                  _ptr_array([outs[i] for i in ii]), n, k, len(ii), m, dy_skinny, _stream()),
                   "skinny_wgrad_batched")
    return outs


def cast_bf16(src, dst):
    """dst[i] = bf16(src[i]) over flat buffers (u3d_cast_bf16)."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    _check(lib().u3d_cast_bf16(_ptr(src), _ptr(dst), C.c_int64(src.numel()), _stream()), "cast_bf16")
    return dst


PERMUTE_DESC_DTYPE = [("src_off", "<i8"), ("dst_off", "<i8"), ("n", "<i4"), ("rows", "<i4"), ("cols", "<i4"), ("reserved", "<i4"),
                      ("stride_k", "<i8"), ("stride_r", "<i8"), ("stride_c", "<i8")]          # = struct u3d_permute_desc


PERMUTE_TILED = True


def permute_plan(descs, device):
    """descs: list of dicts with the u3d_permute_desc fields -> plan for permute_bf16_batched: descriptors whose source is contiguous
    along k go to the LDS-tiled kernel (u3d_permute_bf16_tiled), the rest to the element-wise one."""
    import numpy as np
    arr = np.zeros(len(descs), dtype=PERMUTE_DESC_DTYPE)
    per = int(lib().u3d_permute_block_elems())
    blocks, tiles, max_k = [], [], 1
    for i, d in enumerate(descs):
        for k, v in d.items():
            arr[i][k] = v
        assert d["dst_off"] % 8 == 0
        kk = d["n"] // (d["rows"] * d["cols"])
        if (PERMUTE_TILED and d["stride_k"] == 1 and d["rows"] % 32 == 0 and d["cols"] % 32 == 0 and kk <= 32 and d["src_off"] % 2 == 0
                and kk * d["rows"] * d["cols"] == d["n"] and kk in (d["stride_r"], d["stride_c"])
                and (d["stride_r"] == kk * d["cols"] or d["stride_c"] == kk * d["rows"])):
            tiles += [(i, r, c, 0) for r in range(0, d["rows"], 32) for c in range(0, d["cols"], 32)]
            max_k = max(max_k, kk)
        else:
            blocks += [(i, o) for o in range(0, d["n"], per)]
    descs_dev = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)
    blocks_dev = torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).to(device)
    tiles_dev = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 4).to(device)
    return descs_dev, blocks_dev, len(blocks), tiles_dev, len(tiles), max_k


def permute_bf16_batched(src, dst, plan):
    descs_dev, blocks_dev, nblocks, tiles_dev, ntiles, max_k = plan
    if nblocks:
        _check(lib().u3d_permute_bf16_batched(_ptr(src), _ptr(dst), _ptr(descs_dev), _ptr(blocks_dev), nblocks, _stream()), "permute_bf16_batched")
    if ntiles:
        _check(lib().u3d_permute_bf16_tiled(_ptr(src), _ptr(dst), _ptr(descs_dev), _ptr(tiles_dev), ntiles, max_k, _stream()), "permute_bf16_tiled")
    return dst


def _pc_range6(pc_range):
    return (C.c_float * 6)(*[float(v) for v in pc_range])


def box_decode_fwd(tmp, ref, pc_range, eps=1e-5):
    """tmp [n, code] f32|bf16, ref [n, 3] f32 (sigmoid space) -> decoded codes f32 [n, code] (u3d_box_decode_fwd)."""
    n, code = tmp.shape
    out = torch.empty((n, code), dtype=torch.float32, device=tmp.device)
    _check(lib().u3d_box_decode_fwd(_ptr(tmp), dtype_code(tmp), _ptr(ref), n, code, _pc_range6(pc_range), C.c_float(eps), _ptr(out), _stream()),
           "box_decode_fwd")
    return out


def refine_decode_fwd(tmp, ref_in, ref_s, pc_range, ref_out, ref_sig, eps=1e-5):
    """tmp f32 [n,code], ref_in / ref_s f32 [n,3]; ref_out / ref_sig: preallocated f32 [n,3] views -> decoded codes f32 [n,code]."""
    n, code = tmp.shape
    out = torch.empty((n, code), dtype=torch.float32, device=tmp.device)
    _check(lib().u3d_refine_decode_fwd(_ptr(tmp), _ptr(ref_in), _ptr(ref_s), n, code, _pc_range6(pc_range), C.c_float(eps), _ptr(out),
                                       _ptr(ref_out), _ptr(ref_sig), _stream()), "refine_decode_fwd")
    return out


def box_decode_bwd(tmp, ref, dout, pc_range, eps=1e-5, want_dref=False):
    n, code = tmp.shape
    dtmp = torch.empty_like(tmp)
    dref = torch.empty((n, 3), dtype=torch.float32, device=tmp.device) if want_dref else None
    _check(lib().u3d_box_decode_bwd(_ptr(tmp), dtype_code(tmp), _ptr(ref), _ptr(dout), n, code, _pc_range6(pc_range), C.c_float(eps),
                                    _ptr(dtmp), _ptr(dref), _stream()), "box_decode_bwd")
    return dtmp, dref


def sine_embed_fwd(logits, dim_t, out_dtype):
    """logits f32 [n, nc] -> [n, nc * F] in out_dtype: sin/cos embedding of sigmoid(logits) (u3d_sine_embed_fwd)."""
    n, nc = logits.shape
    F_ = dim_t.numel()
    out = torch.empty((n, nc * F_), dtype=out_dtype, device=logits.device)
    _check(lib().u3d_sine_embed_fwd(_ptr(logits), _ptr(dim_t), n, nc, F_, dtype_code(out), _ptr(out), _stream()), "sine_embed_fwd")
    return out


def sine_embed_bwd(logits, dim_t, dout):
    n, nc = logits.shape
    dl = torch.empty((n, nc), dtype=torch.float32, device=logits.device)
    _check(lib().u3d_sine_embed_bwd(_ptr(logits), _ptr(dim_t), _ptr(dout), dtype_code(dout), n, nc, dim_t.numel(), _ptr(dl), _stream()),
           "sine_embed_bwd")
    return dl


# --------------------------------------------------------------------------------------------------
# fused decoder layer (csrc/decoder.hip, csrc/decoder_bwd.hip)
# --------------------------------------------------------------------------------------------------
_SLOT_CACHE = {}
DT_F32, DT_BF16 = 0, 1                 # include/u3d_hip.h: enum { U3D_F32 = 0, U3D_BF16 = 1 }


def dt_code(dtype):
    """torch dtype of the decoder's element type -> U3D_* code."""
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float32:
        return DT_F32
    raise U3DError(f"fused decoder: unsupported element type {dtype}")


def decoder_layer_slots(m, ncls, code, dtype=torch.bfloat16):
    """(save_off, grad_off): dicts slot name -> byte offset, plus "_total"."""
    key = (m, ncls, code, dtype)
    r = _SLOT_CACHE.get(key)
    if r is None:
        so = (C.c_int64 * (len(DS_NAMES) + 1))()
        go = (C.c_int64 * (len(DG_NAMES) + 1))()
        _check(lib().u3d_decoder_layer_slots_dt(m, ncls, code, dt_code(dtype), so, go), "decoder_layer_slots")
        r = ({**{n: int(so[i]) for i, n in enumerate(DS_NAMES)}, "_total": int(so[len(DS_NAMES)])},
             {**{n: int(go[i]) for i, n in enumerate(DG_NAMES)}, "_total": int(go[len(DG_NAMES)])})
        _SLOT_CACHE[key] = r
    return r


def slot_view(buf, off, rows, cols, dtype):
    """[rows, cols] view of `dtype` at byte offset `off` of the uint8 buffer `buf`."""
    nbytes = rows * cols * torch.empty((), dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype).view(rows, cols)


def decoder_blocks(m, dtype=torch.bfloat16):
    """workgroups of the row-chain kernels = rows of every LayerNorm partial matrix"""
    return int(lib().u3d_decoder_layer_blocks_dt(m, dt_code(dtype)))


def decoder_rows(m, dtype=torch.bfloat16):
    """rows every buffer owned by the fused layer holds: m rounded up to whole row blocks (32 rows in bf16, 16 in f32: the kernels
    never branch on the row)."""
    return decoder_blocks(m, dtype) * (32 if dtype == torch.bfloat16 else 16)


def decoder_layer_fwd(params, dims, x, xc, ref, value_rows, rng, save):
    """xc / value_rows / the slots are in the element type dims.dtype names (bf16 | f32); in f32 mode xc_out IS x_out."""
    m, code, ncls = dims.m, dims.code, dims.ncls
    et = torch.bfloat16 if dims.dtype == DT_BF16 else torch.float32
    assert xc.dtype == et and value_rows.dtype == et and x.dtype == torch.float32
    mp = decoder_rows(m, et)
    dev = x.device
    x_out = torch.empty((mp, 256), dtype=torch.float32, device=dev)
    xc_out = torch.empty((mp, 256), dtype=torch.bfloat16, device=dev) if et == torch.bfloat16 else x_out
    reg = torch.empty((mp, code), dtype=torch.float32, device=dev)
    cls = torch.empty((mp, ncls), dtype=torch.float32, device=dev)
    iou = torch.empty((mp,), dtype=torch.float32, device=dev)
    _check(lib().u3d_decoder_layer_fwd(C.byref(params), C.byref(dims), _ptr(x), _ptr(xc), _ptr(ref), _ptr(value_rows), _ptr(rng),
                                       _ptr(x_out), _ptr(xc_out), _ptr(reg), _ptr(cls), _ptr(iou), _ptr(save), save.numel(), _stream()),
           "decoder_layer_fwd")
    return x_out[:m], xc_out[:m], reg[:m], cls[:m], iou[:m]


def decoder_layer_bwd(params, dims, x, xc, ref, value_rows, rng, xc_out, save, dx_out, dreg, dcls, diou, dvalue, grad):
    """dvalue: f32 [rows, 256] accumulator, or (bf16 kernels only) a bf16 one - packed bf16 atomics, see u3d_declayer_dims.dvalue_bf16."""
    m = dims.m
    dims.dvalue_bf16 = 1 if dvalue.dtype == torch.bfloat16 else 0
    assert dvalue.dtype == torch.float32 or dims.dtype == DT_BF16
    mp = decoder_rows(m, torch.bfloat16 if dims.dtype == DT_BF16 else torch.float32)
    dx = torch.empty((mp, 256), dtype=torch.float32, device=ref.device)
    dref = torch.empty((mp, 3), dtype=torch.float32, device=ref.device) if dims.need_dref else None
    _check(lib().u3d_decoder_layer_bwd(C.byref(params), C.byref(dims), _ptr(x), _ptr(xc), _ptr(ref), _ptr(value_rows), _ptr(rng),
                                       _ptr(xc_out), _ptr(save), _ptr(dx_out), _ptr(dreg), _ptr(dcls), _ptr(diou), _ptr(dx),
                                       _ptr(dvalue), _ptr(dref), _ptr(grad), grad.numel(), _stream()), "decoder_layer_bwd")
    return dx[:m], (None if dref is None else dref[:m])


def mha_fwd(qk, v, nq, p_attn=0.0, layer=0, rng=None):
    """qk [m,512] (q | k), v [m,256] (both bf16 or both f32) -> (o [m,256] same type, lse f32 [m,8] in log2 units)."""
    m = qk.shape[0]
    assert qk.dtype == v.dtype and qk.is_contiguous() and v.is_contiguous()
    o = torch.empty((m, 256), dtype=qk.dtype, device=qk.device)
    lse = torch.empty((m, 8), dtype=torch.float32, device=qk.device)
    _check(lib().u3d_mha_fwd_dt(_ptr(qk), _ptr(v), m, nq, p_attn, layer, _ptr(rng), _ptr(o), _ptr(lse), dt_code(qk.dtype), _stream()),
           "mha_fwd")
    return o, lse


def mha_bwd(qk, v, o, d_o, lse, nq, p_attn=0.0, layer=0, rng=None):
    m = qk.shape[0]
    assert qk.dtype == v.dtype == o.dtype == d_o.dtype
    dqk, dv = torch.empty_like(qk), torch.empty_like(v)
    _check(lib().u3d_mha_bwd_dt(_ptr(qk), _ptr(v), _ptr(o), _ptr(d_o), _ptr(lse), m, nq, p_attn, layer, _ptr(rng), _ptr(dqk), _ptr(dv),
                                dt_code(qk.dtype), _stream()), "mha_bwd")
    return dqk, dv


def wpack(descs_dev, count, max_elems, dtype=torch.bfloat16):
    _check(lib().u3d_wpack(_ptr(descs_dev), count, max_elems, dt_code(dtype), _stream()), "wpack")


def wpack_bf16(descs_dev, count, max_elems):
    wpack(descs_dev, count, max_elems, torch.bfloat16)


def dropout_mask(rng, layer, site, n, p, cols=0):
    """keep mask (bytes) of n elements of dropout site `site`; the attention site (4) takes cols = queries per group."""
    keep = torch.empty((n,), dtype=torch.uint8, device=rng.device)
    _check(lib().u3d_dropout_mask(_ptr(rng), layer, site, n, p, int(cols), _ptr(keep), _stream()), "dropout_mask")
    return keep


# --------------------------------------------------------------------------------------------------
# on-device data path
# --------------------------------------------------------------------------------------------------
AUG_NPARAM = 9


def points_augment(points, scene_off, params, coord, height_dim=-1):
    """In place on points [N,F] f32: per-scene flip -> rotation -> scale -> translation; params f32 [B,9]
    (flip_h, flip_v, sin, cos, angle, scale, tx, ty, tz)."""
    assert points.dtype == torch.float32 and points.is_contiguous() and params.dtype == torch.float32 and params.shape[1] == AUG_NPARAM
    batch = scene_off.numel() - 1
    _check(lib().u3d_points_augment(_ptr(points), _ptr(scene_off), batch, points.shape[0], points.shape[1], _ptr(params.contiguous()),
                                    int(coord), int(height_dim), _stream()), "points_augment")
    return points


def boxes_augment(boxes, gt_off, params, coord):
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape[1] in (7, 9)
    batch = gt_off.numel() - 1
    _check(lib().u3d_boxes_augment(_ptr(boxes), _ptr(gt_off), batch, boxes.shape[0], boxes.shape[1], _ptr(params.contiguous()), int(coord),
                                   _stream()), "boxes_augment")
    return boxes


def boxes_range_filter(boxes, labels, gt_off, bev_range):
    """ObjectRangeFilter in place: (boxes, labels) of every scene compacted to the front of its segment -> count int32 [B]."""
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape[1] in (7, 9)
    assert labels is None or (labels.dtype == torch.int32 and labels.is_contiguous())
    batch = gt_off.numel() - 1
    count = torch.empty((batch,), dtype=torch.int32, device=boxes.device)
    rng = (C.c_float * 4)(*[float(v) for v in bev_range])
    _check(lib().u3d_boxes_range_filter(_ptr(boxes), _ptr(labels), _ptr(gt_off), batch, boxes.shape[1], rng, _ptr(count), _stream()),
           "boxes_range_filter")
    return count


def points_range_filter(points, scene_off, pc_range, out=None):
    """-> (out [N,F] with every scene's survivors compacted to the front of its segment, count int32 [B])."""
    assert points.dtype == torch.float32 and points.is_contiguous()
    batch = scene_off.numel() - 1
    out = torch.empty_like(points) if out is None else out
    count = torch.empty((batch,), dtype=torch.int32, device=points.device)
    rng = (C.c_float * 6)(*[float(v) for v in pc_range])
    _check(lib().u3d_points_range_filter(_ptr(points), _ptr(scene_off), batch, points.shape[1], rng, _ptr(out), _ptr(count), _stream()),
           "points_range_filter")
    return out, count


def point_sample(points, scene_off, count, num_points, seed, want_idx=False):
    """-> out [B*num_points, F] (+ idx int32 [B*num_points]); seed: device int64/uint64 tensor with one element."""
    assert points.dtype == torch.float32 and points.is_contiguous() and seed.numel() == 1 and seed.element_size() == 8
    batch = scene_off.numel() - 1
    out = torch.empty((batch * num_points, points.shape[1]), dtype=torch.float32, device=points.device)
    idx = torch.empty((batch * num_points,), dtype=torch.int32, device=points.device) if want_idx else None
    _check(lib().u3d_point_sample(_ptr(points), _ptr(scene_off), _ptr(count), batch, points.shape[1], int(num_points), _ptr(seed), _ptr(out),
                                  _ptr(idx), _stream()), "point_sample")
    return (out, idx) if want_idx else out
