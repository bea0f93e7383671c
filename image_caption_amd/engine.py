"""Fused whole-sequence decode: the loop of `predict_image_caption_sequence`
(reference recurrent_example.py:477-562) as one enqueue of HIP kernels on a stream."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops

TILE_MODES = {"reference": _lib.TILE_REFERENCE, "repeat": _lib.TILE_REPEAT}
PRECISIONS = {"fp32": _lib.PRECISION_FP32, "bf16": _lib.PRECISION_BF16,
              "fp32_split": _lib.PRECISION_F32_SPLIT}
CELLS = {"lstm": _lib.CELL_LSTM, "gru": _lib.CELL_GRU}

# checkpoint (.npz) parameter name -> icap_weights field
PARAM_TO_FIELD = {
    "enc_kernel": "enc_kernel", "enc_bias": "enc_bias", "embedding": "embedding",
    "init0_kernel": "init_h_kernel", "init0_bias": "init_h_bias",
    "init1_kernel": "init_c_kernel", "init1_bias": "init_c_bias",
    "att_W1_kernel": "att_w1_kernel", "att_W1_bias": "att_w1_bias",
    "att_W2_kernel": "att_w2_kernel", "att_W2_bias": "att_w2_bias",
    "att_V_kernel": "att_v_kernel", "att_V_bias": "att_v_bias",
    "fc3_kernel": "fc3_kernel", "fc3_bias": "fc3_bias",
    "lstm_kernel": "lstm_kernel", "lstm_recurrent_kernel": "lstm_recurrent_kernel",
    "lstm_bias": "lstm_bias", "fc1_kernel": "fc1_kernel", "fc1_bias": "fc1_bias",
    "fc2_kernel": "fc2_kernel", "fc2_bias": "fc2_bias",
}
# the GRU decoder's variables travel in the cell slots of icap_weights; it has one state,
# so the second init-state projection is absent
GRU_PARAM_TO_FIELD = {k: v for k, v in PARAM_TO_FIELD.items()
                      if not k.startswith(("lstm_", "init1_"))}
GRU_PARAM_TO_FIELD.update({"gru_kernel": "lstm_kernel",
                           "gru_recurrent_kernel": "lstm_recurrent_kernel",
                           "gru_bias": "lstm_bias"})


class InvalidDecode(RuntimeError):
    """A decode whose result the device marked invalid (tokens -1, scores NaN)."""


@dataclass
class CaptionConfig:
    """Shapes and decoding constants of one engine (defaults = the reference script:
    embedding_dim 256, units 512, recurrent_example.py:150-152; tokens start=3, eos=4)."""
    B: int
    W: int = 3
    L: int = 64
    F: int = 2048
    D: int = 256
    E: int = 256
    H: int = 512
    U: int = 512
    V: int = 10000
    T: int = 20
    start_id: int = 3
    eos_id: int = 4
    length_penalty_weight: float = 1.0
    reset_state_from_feat: float = 1.0
    tile_mode: str = "reference"
    precision: str = "fp32"
    cell: str = "lstm"            # "gru": the RecurrentDecoder sibling (decoder/recurrent_decoder.py)

    @property
    def N(self):
        return self.B * self.W


class CaptionEngine:
    """Owns an icap_engine handle and the device weights it points at."""

    def __init__(self, config: CaptionConfig, params, device=None):
        if config.tile_mode not in TILE_MODES:
            raise ValueError("tile_mode must be 'reference' or 'repeat'")
        if config.precision not in PRECISIONS:
            raise ValueError("precision must be 'fp32', 'fp32_split' or 'bf16'")
        if config.cell not in CELLS:
            raise ValueError("cell must be 'lstm' or 'gru'")
        self.config = config
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        self._handle = None
        lib = _lib.load()
        # device copies of the weights (kept alive by this object)
        self.weights = {}
        w = _lib.IcapWeights()
        names = GRU_PARAM_TO_FIELD if config.cell == "gru" else PARAM_TO_FIELD
        for pname, field in names.items():
            if pname not in params:
                raise KeyError("missing parameter %r" % pname)
            t = params[pname]
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            self.weights[field] = t
            setattr(w, field, t.data_ptr())
        self._check_shapes()
        c = _lib.IcapConfig(B=config.B, W=config.W, L=config.L, F=config.F, D=config.D,
                            E=config.E, H=config.H, U=config.U, V=config.V, T=config.T,
                            start_id=config.start_id, eos_id=config.eos_id,
                            tile_mode=TILE_MODES[config.tile_mode],
                            precision=PRECISIONS[config.precision],
                            length_penalty_weight=float(config.length_penalty_weight),
                            reset_state_from_feat=float(config.reset_state_from_feat),
                            cell=CELLS[config.cell])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.icap_engine_create(C.byref(c), C.byref(w), C.byref(handle)))
        self._handle = handle
        self._versions = self._weight_versions()

    def _weight_versions(self):
        return {k: t._version for k, t in self.weights.items()}

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def refresh_weights(self):
        """Recomputes what the engine derives from the weights (token part of the gates, GRU
        re-layout, packed bf16 / split copies, the two scalar biases) after their values were
        changed in place.  Waits for the current stream."""
        with torch.cuda.device(self.device):
            _lib.call("icap_engine_refresh_weights", self._handle, self._stream())
        self._versions = self._weight_versions()

    def refresh_if_stale(self):
        """The engine points at the caller's weight tensors (no copy when they already are
        float32 on the device).  Every decode compares their torch version counters with the
        ones seen at the last refresh, so an optimizer step or `kernel.copy_()` between two
        decodes is picked up; writes that bypass torch need an explicit refresh_weights()."""
        if self._weight_versions() != self._versions:
            self.refresh_weights()
            return True
        return False

    def _check_shapes(self):
        c, w = self.config, self.weights
        expect = {
            "enc_kernel": (c.F, c.D), "enc_bias": (c.D,), "embedding": (c.V, c.E),
            "init_h_kernel": (c.D, c.H), "init_h_bias": (c.H,), "init_c_kernel": (c.D, c.H),
            "init_c_bias": (c.H,), "att_w1_kernel": (c.D, c.U), "att_w1_bias": (c.U,),
            "att_w2_kernel": (c.H, c.U), "att_w2_bias": (c.U,), "att_v_kernel": (c.U, 1),
            "att_v_bias": (1,), "fc3_kernel": (c.H, 1), "fc3_bias": (1,),
            "lstm_kernel": (c.D + c.E, 4 * c.H), "lstm_recurrent_kernel": (c.H, 4 * c.H),
            "lstm_bias": (4 * c.H,), "fc1_kernel": (c.H, c.H), "fc1_bias": (c.H,),
            "fc2_kernel": (c.H, c.V), "fc2_bias": (c.V,)}
        if c.cell == "gru":
            del expect["init_c_kernel"], expect["init_c_bias"]
            expect.update({"lstm_kernel": (c.D + c.E, 3 * c.H),
                           "lstm_recurrent_kernel": (c.H, 3 * c.H), "lstm_bias": (2, 3 * c.H)})
        for k, shape in expect.items():
            if tuple(w[k].shape) != shape:
                raise ValueError("weight %s has shape %s, expected %s" % (k, tuple(w[k].shape), shape))

    def close(self):
        if self._handle is not None:
            _lib.load().icap_engine_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(_lib.load().icap_engine_workspace_bytes(self._handle))

    def decode(self, img, out=None, scores=None, check=False):
        """img [B,L,F] float32 on the engine's device -> (tokens [B,T] int32, scores [B]).
        check=True waits for the decode and raises InvalidDecode if the hand-over inside the
        merged launch ran out of its bounded wait (see decode_invalid)."""
        c = self.config
        if tuple(img.shape) != (c.B, c.L, c.F):
            raise ValueError("img must have shape %s" % ((c.B, c.L, c.F),))
        if out is None:
            out = torch.empty(c.B, c.T, dtype=torch.int32, device=self.device)
        if scores is None:
            scores = torch.empty(c.B, dtype=torch.float32, device=self.device)
        self._same_device(img=img, out=out, scores=scores)
        self.refresh_if_stale()
        ops._call("icap_decode", self._handle, ops._chk(img, torch.float32, "img"),
                  ops._chk(out, torch.int32, "out"), ops._chk(scores, torch.float32, "scores"),
                  ops.current_stream())
        if check:
            self.raise_if_invalid()
        return out, scores

    def decode_invalid(self):
        """Waits for the current stream and reports whether the last decode was invalidated on
        the device: a bounded wait of the in-launch hand-over ran out, every token is -1 and
        every score NaN (never observed in normal operation; icap_engine_decode_status)."""
        flag = C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.call("icap_engine_decode_status", self._handle, C.byref(flag), self._stream())
        return bool(flag.value)

    def raise_if_invalid(self):
        if self.decode_invalid():
            raise InvalidDecode("the decode was invalidated on the device: a wait inside the merged "
                                "launch ran out; its tokens are -1")

    def _same_device(self, **tensors):
        for name, t in tensors.items():
            if t is not None and isinstance(t, torch.Tensor) and t.is_cuda and t.device != self.device:
                raise ValueError("%s lives on %s, the engine on %s" % (name, t.device, self.device))

    def search_baseline(self, features, seq_len=None, pad_token=0, out=None, scores=None):
        """The baseline beam searcher's loop (reference utils/agent_utils.py:152-262) on
        encoded features [B,L,D]: returns (tokens [B,T] int32 without the start column,
        scores [B]).  seq_len [B] int32: target lengths for the forced eos / pad masks."""
        c = self.config
        if tuple(features.shape) != (c.B, c.L, c.D):
            raise ValueError("features must have shape %s" % ((c.B, c.L, c.D),))
        if out is None:
            out = torch.empty(c.B, c.T, dtype=torch.int32, device=self.device)
        if scores is None:
            scores = torch.empty(c.B, dtype=torch.float32, device=self.device)
        if seq_len is not None and tuple(seq_len.shape) != (c.B,):
            raise ValueError("seq_len must have shape (%d,)" % c.B)
        self._same_device(features=features, seq_len=seq_len, out=out, scores=scores)
        self.refresh_if_stale()
        ops._call("icap_search_baseline", self._handle,
                  ops._chk(features, torch.float32, "features"),
                  ops._opt(seq_len, torch.int32, "seq_len"), int(pad_token),
                  ops._chk(out, torch.int32, "out"), ops._chk(scores, torch.float32, "scores"),
                  ops.current_stream())
        return out, scores

    def trace(self):
        """Per-step records of the last decode as tensors [T,B,W]: ids, parents (int32),
        scores, log_probs (float32)."""
        c = self.config
        shape = (c.T, c.B, c.W)
        out = {"ids": torch.empty(shape, dtype=torch.int32, device=self.device),
               "parents": torch.empty(shape, dtype=torch.int32, device=self.device),
               "scores": torch.empty(shape, dtype=torch.float32, device=self.device),
               "log_probs": torch.empty(shape, dtype=torch.float32, device=self.device)}
        with torch.cuda.device(self.device):
            _lib.call("icap_engine_trace", self._handle, out["ids"].data_ptr(),
                      out["parents"].data_ptr(), out["scores"].data_ptr(),
                      out["log_probs"].data_ptr(), self._stream())
        return out

    @property
    def fused_vocab(self):
        """True when the vocab projection and the beam step run fused (no logits buffer)."""
        return bool(_lib.load().icap_engine_fused_vocab(self._handle))

    def set_fused_vocab(self, on):
        """Select the fused (default where served) or the unfused projection + beam step."""
        _lib.call("icap_engine_set_fused_vocab", self._handle, 1 if on else 0)

    @property
    def merged_launch(self):
        """True when the next step's query + attention ride in the vocab projection's launch."""
        return bool(_lib.load().icap_engine_merged_launch(self._handle))

    @property
    def fold_fc(self):
        """True when the folded fc1 . fc2 kernel projects the cell output."""
        return bool(_lib.load().icap_engine_fold_fc(self._handle))

    @property
    def att_table(self):
        """True when the attention scores use the per-decode exp(2 PF) table."""
        return bool(_lib.load().icap_engine_att_table(self._handle))

    @property
    def small_loop(self):
        """True when greedy decodes run their whole step loop as one launch of resident workgroups
        (ICAP_SMALL_LOOP=1 when the engine was created, supported shapes only)."""
        return bool(_lib.load().icap_engine_small_loop(self._handle))

    def set_fold_fc(self, on):
        """True (default in fp32 / split mode): project the cell output with the folded
        fc1 . fc2 kernel; False: the reference's two products."""
        _lib.call("icap_engine_set_fold_fc", self._handle, 1 if on else 0)

    def set_overlap(self, on):
        """False keeps every launch on the caller's stream (no side-stream attention)."""
        _lib.call("icap_engine_set_overlap", self._handle, 1 if on else 0)

    def set_timing(self, kernels, every=1):
        """kernels: True (all classes), False/None (off) or an iterable of class names
        from _lib.K_NAMES whose launches are bracketed by HIP events; every: bracket them in
        every n-th decode call only."""
        _lib.call("icap_engine_set_timing_every", self._handle, int(every))
        if kernels is True:
            mask = (1 << len(_lib.K_NAMES)) - 1
        elif not kernels:
            mask = 0
        else:
            mask = 0
            for k in kernels:
                mask |= 1 << _lib.K_NAMES.index(k)
        _lib.call("icap_engine_set_timing", self._handle, mask)

    def kernel_times(self):
        """{kernel class: (total ms, launches)} since timing was enabled."""
        n = len(_lib.K_NAMES)
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        _lib.call("icap_engine_kernel_times", self._handle, ms, cnt)
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.K_NAMES)}


def params_from_modules(encoder, decoder):
    """Collects the device weights of an MlpEncoder + LstmDecoder (or RecurrentDecoder) pair
    under the engine's parameter names."""
    from .decoder.embedding import EmbeddingLayer
    emb = decoder.embedding_layer
    if not isinstance(emb, EmbeddingLayer):
        raise TypeError("sequence_beam_search needs decoder.embedding_layer to be an "
                        "EmbeddingLayer so the engine can read the table")
    att = decoder.attention
    if hasattr(decoder, "gru"):
        cell = {"gru_kernel": decoder.gru.kernel,
                "gru_recurrent_kernel": decoder.gru.recurrent_kernel, "gru_bias": decoder.gru.bias}
    else:
        cell = {"init1_kernel": decoder.feat_proj_state_vec[1].fc.kernel,
                "init1_bias": decoder.feat_proj_state_vec[1].fc.bias,
                "lstm_kernel": decoder.lstm.kernel,
                "lstm_recurrent_kernel": decoder.lstm.recurrent_kernel,
                "lstm_bias": decoder.lstm.bias}
    return {
        **cell,
        "enc_kernel": encoder.fc.kernel, "enc_bias": encoder.fc.bias, "embedding": emb.table,
        "init0_kernel": decoder.feat_proj_state_vec[0].fc.kernel,
        "init0_bias": decoder.feat_proj_state_vec[0].fc.bias,
        "att_W1_kernel": att.W1.kernel, "att_W1_bias": att.W1.bias,
        "att_W2_kernel": att.W2.kernel, "att_W2_bias": att.W2.bias,
        "att_V_kernel": att.V.kernel, "att_V_bias": att.V.bias,
        "fc3_kernel": decoder.fc3.kernel, "fc3_bias": decoder.fc3.bias,
        "fc1_kernel": decoder.fc1.kernel, "fc1_bias": decoder.fc1.bias,
        "fc2_kernel": decoder.fc2.kernel, "fc2_bias": decoder.fc2.bias,
    }


_engine_cache = {}


def sequence_beam_search(encoder, decoder, img, beam_width, max_len, start_id, eos_id,
                         length_penalty_weight=1.0, tile_mode="reference", precision="fp32",
                         return_scores=False):
    """Whole-sequence beam search with the semantics of the reference driver loop
    `predict_image_caption_sequence` (recurrent_example.py:477-562).

    img [B,L,F] -> int32 tokens [B,T] (and best-beam scores [B] if asked).  The engine
    (workspace + weight pointers) is cached per (modules, shapes, constants, weight storage);
    weight values changed in place between two calls (an optimizer step, `kernel.copy_()`) are
    noticed through the tensors' version counters and the derived buffers recomputed."""
    B, L, F = img.shape
    if hasattr(encoder, "_is_relu") and not encoder._is_relu():
        raise NotImplementedError("sequence_beam_search: the fused engine's encoder is Dense + ReLU; "
                                  "run another activation through the module-level loop")
    if not encoder.fc.built:
        encoder.fc.build(F, img.device)
    decoder._build(encoder.fc.units, img.device)
    p = params_from_modules(encoder, decoder)
    cfg = CaptionConfig(B=B, W=beam_width, L=L, F=F, D=encoder.fc.units,
                        E=decoder.embedding_dim, H=decoder.units, U=decoder.attention.units,
                        V=decoder.vocab_size, T=max_len, start_id=start_id, eos_id=eos_id,
                        length_penalty_weight=length_penalty_weight,
                        reset_state_from_feat=float(decoder.reset_state_from_feat),
                        tile_mode=tile_mode, precision=precision,
                        cell="gru" if hasattr(decoder, "gru") else "lstm")
    key = (id(encoder), id(decoder), tuple(sorted(cfg.__dict__.items())),
           tuple((k, v.data_ptr()) for k, v in sorted(p.items())), str(img.device))
    eng = _engine_cache.get(key)
    if eng is None:
        eng = CaptionEngine(cfg, p, device=img.device)
        _engine_cache.clear()            # one live engine per call site keeps HBM use flat
        _engine_cache[key] = eng
    tokens, scores = eng.decode(img.contiguous())
    return (tokens, scores) if return_scores else tokens
