"""ctypes binding of libicap.so (the C ABI declared in include/icap.h).

The HIP library is the only compute path of this package: if it is missing the
import fails loudly instead of falling back to anything else."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicap.so")

ICAP_OK = 0
ACT_NONE, ACT_RELU = 0, 1
TILE_REFERENCE, TILE_REPEAT = 0, 1
PRECISION_FP32, PRECISION_BF16, PRECISION_F32_SPLIT = 0, 1, 2
CELL_LSTM, CELL_GRU = 0, 1
MAX_BEAM_WIDTH = 16
BEAM_STEP_STATIC_MAX_ROWS = 16384
K_NAMES = ("encode", "feat_proj", "init_state", "attn_query", "attention", "lstm", "fc1",
           "vocab", "beam", "backtrack")


class IcapError(RuntimeError):
    """A libicap entry point returned a negative status."""

    def __init__(self, code, message):
        super().__init__("icap error %d: %s" % (code, message))
        self.code = code


class IcapConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "W", "L", "F", "D", "E", "H", "U", "V", "T",
                                         "start_id", "eos_id", "tile_mode", "precision")] + [
        ("length_penalty_weight", C.c_double),
        ("reset_state_from_feat", C.c_float),
        ("cell", C.c_int32)]


WEIGHT_FIELDS = ("enc_kernel", "enc_bias", "embedding", "init_h_kernel", "init_h_bias",
                 "init_c_kernel", "init_c_bias", "att_w1_kernel", "att_w1_bias",
                 "att_w2_kernel", "att_w2_bias", "att_v_kernel", "att_v_bias", "fc3_kernel",
                 "fc3_bias", "lstm_kernel", "lstm_recurrent_kernel", "lstm_bias", "fc1_kernel",
                 "fc1_bias", "fc2_kernel", "fc2_bias")


class IcapWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in WEIGHT_FIELDS]


TMASK_AS_WRITTEN, TMASK_CAUSAL = 0, 1

TLAYER_FIELDS = ("self_ln_gamma", "self_ln_beta", "self_q_kernel", "self_q_bias", "self_k_kernel",
                 "self_k_bias", "self_v_kernel", "self_v_bias", "self_o_kernel", "self_o_bias",
                 "cross_ln_gamma", "cross_ln_beta", "cross_q_kernel", "cross_q_bias",
                 "cross_k_kernel", "cross_k_bias", "cross_v_kernel", "cross_v_bias",
                 "cross_o_kernel", "cross_o_bias", "ffn_ln_gamma", "ffn_ln_beta",
                 "ffn_filter_kernel", "ffn_filter_bias", "ffn_output_kernel", "ffn_output_bias")
TWEIGHT_FIELDS = ("enc_kernel", "enc_bias", "embedding", "out_ln_gamma", "out_ln_beta",
                  "fc2_kernel", "fc2_bias")
TK_NAMES = ("encode", "embed", "layernorm", "proj", "self_att", "cross_att", "ffn", "vocab", "beam",
            "backtrack")


class IcapTConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "W", "L", "F", "D", "hidden", "heads", "key_dim",
                                         "filter", "layers", "V", "T", "start_id", "eos_id",
                                         "tile_mode", "mask_mode", "kv_from_normed", "precision")] + [
        ("length_penalty_weight", C.c_double)]


class IcapTLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TLAYER_FIELDS]


class IcapTWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TWEIGHT_FIELDS] + [("layers", C.POINTER(IcapTLayer))]


_P, _I, _F, _D = C.c_void_p, C.c_int, C.c_float, C.c_double

# name -> argtypes; every entry point of include/icap.h
SIGNATURES = {
    "icap_version": ([], C.c_int),
    "icap_last_error": ([C.c_char_p, C.c_size_t], C.c_size_t),
    "icap_dense_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "icap_pool_features_f32": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_expand_rows_f32": ([_P, _P, _I, _I, _I, _F, _I, _P], C.c_int),
    "icap_embedding_f32": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_attention_f32": ([_P, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P,
                            _I, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_attention_act_f32": ([_P, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P,
                                _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_lstm_step_f32": ([_P] * 12 + [_I] * 5 + [_P], C.c_int),
    "icap_gru_pack_f32": ([_P] * 6 + [_I] * 2 + [_P], C.c_int),
    "icap_gru_step_bf16": ([_P] * 7 + [_I] * 4 + [_P], C.c_int),
    "icap_gru_step_f32": ([_P] * 10 + [_I] * 5 + [_P], C.c_int),
    "icap_beam_step_f32": ([_P, _P, _P, _P, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P,
                            _I, _I, _I, _P, C.c_size_t, _P], C.c_int),
    "icap_beam_step_rows_f32": ([_P, _P, _P, _P, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P,
                                 _I, _I, _I, _I, _P, C.c_size_t, _P], C.c_int),
    "icap_beam_step_workspace_bytes": ([_I, _I], C.c_size_t),
    "icap_beam_step_static_f32": ([_P, _P, _P, _P, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P,
                                   _I, _I, _I, _P], C.c_int),
    "icap_sample_categorical_f32": ([_P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P], C.c_int),
    "icap_backtrack": ([_P] * 7 + [_I, _I, _I, _P], C.c_int),
    "icap_packed_weight_bytes": ([_I, _I], C.c_size_t),
    "icap_pack_weight_bf16": ([_P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_dense_bf16": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "icap_lstm_step_bf16": ([_P] * 9 + [_I, _I, _I, _I, _P], C.c_int),
    "icap_packed_split_bytes": ([_I, _I], C.c_size_t),
    "icap_pack_weight_split": ([_P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_dense_split": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "icap_dense_topk_split": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "icap_lstm_step_split": ([_P] * 9 + [_I, _I, _I, _I, _P], C.c_int),
    "icap_dense_topk_half_tiles": ([_I], C.c_int),
    "icap_dense_topk_record_floats": ([_I], C.c_int),
    "icap_dense_topk_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "icap_beam_step_from_records_f32": ([_P, _P, _I, _I, _P, _P, _P, _I, _I, _D, _P, _P, _P, _P,
                                         _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_beam_step_from_records_wide_f32": ([_P, _P, _I, _I, _P, _P, _P, _I, _I, _D, _P, _P, _P,
                                              _P, _P, _P, _I, _I, _I, _P, _I, _P], C.c_int),
    "icap_beam_scores_f32": ([_P, _P, _P, _P, _I, _D, _I, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_beam_gather_f32": ([_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "icap_mask_probs_f32": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_hyp_score_f32": ([_P, _P, _D, _P, C.c_longlong, _P], C.c_int),
    "icap_batch_gather_f32": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_engine_create": ([C.POINTER(IcapConfig), C.POINTER(IcapWeights), C.POINTER(_P)], C.c_int),
    "icap_engine_destroy": ([_P], C.c_int),
    "icap_engine_refresh_weights": ([_P, _P], C.c_int),
    "icap_engine_workspace_bytes": ([_P], C.c_size_t),
    "icap_decode": ([_P, _P, _P, _P, _P], C.c_int),
    "icap_position_table_f32": ([_P, _I, _I, _P], C.c_int),
    "icap_embed_pos_f32": ([_P, _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "icap_add_f32": ([_P, _P, C.c_longlong, C.c_longlong, _P], C.c_int),
    "icap_residual_layernorm_f32": ([_P, _P, _P, _P, _P, _I, _I, _F, _P], C.c_int),
    "icap_mha_self_step_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_mha_cross_step_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "icap_mha_scores_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_mha_combine_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "icap_tengine_create": ([_P, _P, _P], C.c_int),
    "icap_tengine_destroy": ([_P], C.c_int),
    "icap_tengine_refresh_weights": ([_P, _P], C.c_int),
    "icap_tengine_workspace_bytes": ([_P], C.c_size_t),
    "icap_tdecode": ([_P, _P, _P, _P, _P], C.c_int),
    "icap_tengine_trace": ([_P, _P, _P, _P, _P, _P], C.c_int),
    "icap_tengine_last_output": ([_P, _P, _P], C.c_int),
    "icap_tengine_set_timing": ([_P, C.c_uint], C.c_int),
    "icap_tengine_kernel_times": ([_P, _P, _P], C.c_int),
    "icap_search_baseline": ([_P, _P, _P, _I, _P, _P, _P], C.c_int),
    "icap_logit_masks_f32": ([_P, _I, _I, _I, _P, _I, _I, _I, _P], C.c_int),
    "icap_engine_trace": ([_P, _P, _P, _P, _P, _P], C.c_int),
    "icap_engine_set_timing": ([_P, C.c_uint], C.c_int),
    "icap_engine_set_timing_every": ([_P, C.c_int], C.c_int),
    "icap_engine_set_overlap": ([_P, _I], C.c_int),
    "icap_engine_set_fold_fc": ([_P, _I], C.c_int),
    "icap_engine_merged_launch": ([_P], C.c_int),
    "icap_engine_fold_fc": ([_P], C.c_int),
    "icap_engine_att_table": ([_P], C.c_int),
    "icap_engine_small_loop": ([_P], C.c_int),
    "icap_engine_decode_status": ([_P, C.POINTER(C.c_int32), _P], C.c_int),
    "icap_engine_debug_sync_state": ([_P, _P, C.POINTER(C.c_int32)], C.c_int),
    "icap_engine_set_fused_vocab": ([_P, _I], C.c_int),
    "icap_engine_fused_vocab": ([_P], C.c_int),
    "icap_engine_kernel_times": ([_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)], C.c_int),
}

_lib = None


def load():
    """Loads libicap.so once and declares every signature."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same soname as /opt/rocm's); importing it first
    # makes libicap.so bind to that one, so torch's device pointers and streams are valid
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "image_caption_amd: %s is missing - the HIP extension is the only compute path; "
            "build it with `make -C image_caption_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def last_error():
    lib = load()
    buf = C.create_string_buffer(1024)
    lib.icap_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc != ICAP_OK:
        raise IcapError(rc, last_error())


def call(name, *args):
    check(getattr(load(), name)(*args))
