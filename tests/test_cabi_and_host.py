"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol
that include/icap.h declares (no compute calls), argument validation that happens on
the host, and the reference-shaped module surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "icap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icap_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for name in ("icap_dense_f32", "icap_attention_f32", "icap_lstm_step_f32",
                 "icap_beam_step_f32", "icap_backtrack", "icap_engine_create", "icap_decode",
                 "icap_engine_destroy", "icap_last_error", "icap_version"):
        assert name in syms


def test_library_loads_and_exports_every_declared_symbol():
    from image_caption_amd import _lib
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), "libicap.so does not export %s" % name
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
    assert set(_lib.SIGNATURES) == set(_declared_symbols())
    assert lib.icap_version() >= 100


def test_last_error_is_reported_without_a_gpu():
    from image_caption_amd import _lib
    lib = _lib.load()
    # null pointers are rejected on the host before any HIP call
    rc = lib.icap_dense_f32(None, None, None, None, 4, 4, 4, 0, None)
    assert rc == -1
    assert "null" in _lib.last_error()
    rc = lib.icap_backtrack(None, None, None, None, None, None, None, 1, 1, 1, None)
    assert rc == -1
    assert lib.icap_beam_step_workspace_bytes(512, 3) >= 512 * 3 * 8
    with pytest.raises(_lib.IcapError):
        _lib.call("icap_engine_set_timing", None, 0)


def test_struct_layouts_match_the_header():
    from image_caption_amd import _lib
    # 14 int32, one double (8-aligned), float, int32
    assert ctypes.sizeof(_lib.IcapConfig) == 14 * 4 + 8 + 4 + 4
    assert _lib.IcapConfig.length_penalty_weight.offset == 56
    assert ctypes.sizeof(_lib.IcapWeights) == 22 * ctypes.sizeof(ctypes.c_void_p)
    text = open(os.path.join(ROOT, "include", "icap.h")).read()
    body = text[text.index("typedef struct icap_weights"):text.index("} icap_weights;")]
    fields = re.findall(r"const float\*\s*([a-z0-9_]+);", body)
    assert tuple(fields) == _lib.WEIGHT_FIELDS


def test_ops_reject_cpu_tensors_and_wrong_dtypes():
    from image_caption_amd import ops
    x = torch.zeros(4, 8)
    w = torch.zeros(8, 4)
    with pytest.raises(ValueError):
        ops.dense(x, w)                      # not on the GPU
    with pytest.raises(ValueError):
        ops.dense(x, torch.zeros(9, 4))      # shape mismatch is caught first


def test_module_surface_mirrors_the_reference():
    from image_caption_amd.attention.attention import BahdanauAttention
    from image_caption_amd.beam_search import beam_search as bs
    from image_caption_amd.decoder.decoder import Decoder
    from image_caption_amd.decoder.embedding import EmbeddingLayer
    from image_caption_amd.decoder.init_state import FeatureProjState
    from image_caption_amd.decoder.lstm_recurrent_decoder import LstmDecoder
    from image_caption_amd.encoder.encoder import Encoder
    from image_caption_amd.encoder.mlp_encoder import MlpEncoder
    assert issubclass(MlpEncoder, Encoder) and issubclass(LstmDecoder, Decoder)
    for name in ("BeamSearchState", "BeamSearchStepOutput", "BeamSearchConfig",
                 "create_initial_beam_state", "create_initial_beam_state_batch",
                 "combine_beam_state_batch", "choose_top_k", "beam_search_step",
                 "beam_search_step_batch", "loc_optimal_beam_path", "loc_optimal_beam_path_batch",
                 "sequence_beam_search", "mask_probs", "hyp_score", "length_penalty",
                 "batch_gather"):
        assert hasattr(bs, name)
    assert bs.BeamSearchConfig._fields == ("beam_width", "vocab_size", "eos_token",
                                           "length_penalty_weight", "choose_successors_fn")
    assert bs.BeamSearchState._fields == ("log_probs", "finished", "lengths")
    assert bs.BeamSearchStepOutput._fields == ("scores", "predicted_ids", "beam_parent_ids")
    emb = EmbeddingLayer(11, 4, device="cpu")
    dec = LstmDecoder(emb, 4, 6, 11, reset_state_from_feat=1, name="d", device="cpu")
    assert dec.num_cells == 2 and dec.vocab_size == 11 and dec.data_type == torch.float32
    cp = dec.copy()
    assert isinstance(cp, LstmDecoder) and cp is not dec and cp.units == 6
    assert cp.reset_state_from_feat == 1 and cp.embedding_layer is emb
    # the mandatory `training` keyword (KeyError like the reference) is checked before any compute
    enc = MlpEncoder(4, device="cpu")
    with pytest.raises(KeyError):
        enc(torch.zeros(1, 2, 3))
    with pytest.raises(KeyError):
        dec((torch.zeros(1, 1, dtype=torch.int32), {"features": torch.zeros(1, 2, 4)},
             [torch.zeros(1, 6), torch.zeros(1, 6)]))
    st = dec.reset_state(torch.zeros(3, 1), {})
    assert len(st) == 2 and tuple(st[0].shape) == (3, 6) and float(st[0].abs().sum()) == 0.0
    assert isinstance(BahdanauAttention(6, device="cpu"), BahdanauAttention)
    assert FeatureProjState(6, "x", device="cpu").fc.units == 6
    s0 = bs.create_initial_beam_state_batch(2, 3, device="cpu")
    assert s0.finished.dtype == torch.bool and s0.lengths.dtype == torch.int32
    both = bs.combine_beam_state_batch(s0, s0)
    assert tuple(both.log_probs.shape) == (2, 6)


def test_keras_initialisers():
    from image_caption_amd.decoder.lstm_recurrent_decoder import LstmCellWeights
    from image_caption_amd.layers import Dense, Embedding
    d = Dense(7, device="cpu")
    d.build(5)
    lim = np.sqrt(6.0 / 12)
    assert tuple(d.kernel.shape) == (5, 7) and float(d.kernel.abs().max()) <= lim
    assert float(d.bias.abs().sum()) == 0.0
    e = Embedding(13, 3, device="cpu")
    e.build()
    assert float(e.table.abs().max()) <= 0.05
    c = LstmCellWeights(4, device="cpu")
    c.build(6)
    assert tuple(c.kernel.shape) == (6, 16) and tuple(c.recurrent_kernel.shape) == (4, 16)
    assert c.bias.tolist() == [0.0] * 4 + [1.0] * 4 + [0.0] * 8     # unit_forget_bias


def test_engine_config_validation_on_host():
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    with pytest.raises(ValueError):
        CaptionEngine(CaptionConfig(B=1, tile_mode="nope"), {})
    with pytest.raises(ValueError):
        CaptionEngine(CaptionConfig(B=1, precision="fp8"), {})
    assert CaptionConfig(B=4, W=3).N == 12


def test_synthetic_generators_match_the_oracle_generators():
    from image_caption_amd import synthetic as sy
    from oracle import caption_oracle as co
    a = co.make_params(3, 24, 8, 8, 12, 37)
    b = sy.make_params(3, 24, 8, 8, 12, 37)
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    assert np.array_equal(co.make_images(1, 2, 3, 4), sy.make_images(1, 2, 3, 4))


def test_weight_container_roundtrip(tmp_path):
    from image_caption_amd import synthetic as sy
    p = sy.make_params(0, 8, 4, 4, 6, 11)
    path = str(tmp_path / "w.npz")
    sy.save_params(path, p)
    q = sy.load_params(path)
    assert q.keys() == p.keys() and all(np.array_equal(p[k], q[k]) for k in p)


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: no file of the package (Python or HIP) names it,
    and importing the whole package leaves it unloaded."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "image_caption_amd")
    hits = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(d, f)) as fh:
                    if "oracle" in fh.read():
                        hits.append(os.path.join(d, f))
    assert hits == []
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import image_caption_amd, image_caption_amd.engine, image_caption_amd.distributed\n"
            "from image_caption_amd.beam_search import beam_search\n"
            "from image_caption_amd.decoder import lstm_recurrent_decoder\n"
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)\n") % ROOT
    subprocess.run([sys.executable, "-c", code], check=True, timeout=120)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No CPU fallback: without libicap.so the loader raises instead of computing elsewhere."""
    from image_caption_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libicap.so"))
    with pytest.raises(ImportError):
        _lib.load()
