"""Fused whole-sequence decode over the transformer decoder (reference
decoder/transform_decoder.py): the loop of `predict_image_caption_sequence`
(recurrent_example.py:477-562) with the decoder swapped, as one enqueue of HIP kernels."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops
from .engine import TILE_MODES

MASK_MODES = {"as_written": _lib.TMASK_AS_WRITTEN, "causal": _lib.TMASK_CAUSAL}


@dataclass
class TransformerConfig:
    """Shapes of one engine.  As the reference is written, `MultiHeadAttention(hidden_size,
    num_heads)` binds positionally to (num_heads=hidden_size, key_dim=num_heads)
    (transform_decoder.py:65-72): `heads` = hidden_size and `key_dim` = params["num_heads"]
    reproduce it (the defaults); heads=8, key_dim=64 would be the conventional layout."""
    B: int
    W: int = 3
    L: int = 64
    F: int = 2048
    D: int = 512
    hidden: int = 512
    heads: int = 512
    key_dim: int = 8
    filter: int = 2048
    layers: int = 6
    V: int = 10000
    T: int = 20
    start_id: int = 3
    eos_id: int = 4
    length_penalty_weight: float = 1.0
    tile_mode: str = "reference"
    mask_mode: str = "as_written"
    kv_from_normed: bool = False

    @property
    def N(self):
        return self.B * self.W


def layer_param_names(l):
    return {f: "l%d_%s" % (l, f) for f in _lib.TLAYER_FIELDS}


class TransformerEngine:
    """Owns an icap_tengine handle and the device weights it points at.  `params` uses the
    names `enc_kernel`, `enc_bias`, `t_embedding`, `t_fc2_kernel`, `t_fc2_bias`,
    `out_ln_gamma`, `out_ln_beta` and `l<i>_<field>` for every field of icap_tlayer."""

    TOP = {"enc_kernel": "enc_kernel", "enc_bias": "enc_bias", "t_embedding": "embedding",
           "out_ln_gamma": "out_ln_gamma", "out_ln_beta": "out_ln_beta",
           "t_fc2_kernel": "fc2_kernel", "t_fc2_bias": "fc2_bias"}

    def __init__(self, config: TransformerConfig, params, device=None):
        if config.tile_mode not in TILE_MODES:
            raise ValueError("tile_mode must be 'reference' or 'repeat'")
        if config.mask_mode not in MASK_MODES:
            raise ValueError("mask_mode must be 'as_written' or 'causal'")
        self.config = config
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        self._handle = None
        lib = _lib.load()
        self.weights = {}

        def dev(name):
            if name not in params:
                raise KeyError("missing parameter %r" % name)
            t = params[name]
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            self.weights[name] = t
            return t

        c = config
        HD = c.heads * c.key_dim
        w = _lib.IcapTWeights()
        top_shapes = {"enc_kernel": (c.F, c.D), "enc_bias": (c.D,), "t_embedding": (c.V, c.hidden),
                      "out_ln_gamma": (c.hidden,), "out_ln_beta": (c.hidden,),
                      "t_fc2_kernel": (c.hidden, c.V), "t_fc2_bias": (c.V,)}
        for pname, field in self.TOP.items():
            t = dev(pname)
            self._expect(pname, t, top_shapes[pname])
            setattr(w, field, t.data_ptr())
        layer_shapes = {"ffn_ln_gamma": (c.hidden,), "ffn_ln_beta": (c.hidden,),
                        "ffn_filter_kernel": (c.hidden, c.filter), "ffn_filter_bias": (c.filter,),
                        "ffn_output_kernel": (c.filter, c.hidden), "ffn_output_bias": (c.hidden,)}
        for att, d_kv in (("self_", c.hidden), ("cross_", c.D)):
            layer_shapes.update({
                att + "ln_gamma": (c.hidden,), att + "ln_beta": (c.hidden,),
                att + "q_kernel": (c.hidden, HD), att + "k_kernel": (d_kv, HD),
                att + "v_kernel": (d_kv, HD), att + "o_kernel": (HD, c.hidden),
                att + "q_bias": (HD,), att + "k_bias": (HD,), att + "v_bias": (HD,),
                att + "o_bias": (c.hidden,)})
        self._layers = (_lib.IcapTLayer * c.layers)()
        for l in range(c.layers):
            for field, pname in layer_param_names(l).items():
                t = dev(pname)
                self._expect(pname, t, layer_shapes[field])
                setattr(self._layers[l], field, t.data_ptr())
        w.layers = C.cast(self._layers, C.POINTER(_lib.IcapTLayer))
        cc = _lib.IcapTConfig(B=c.B, W=c.W, L=c.L, F=c.F, D=c.D, hidden=c.hidden, heads=c.heads,
                              key_dim=c.key_dim, filter=c.filter, layers=c.layers, V=c.V, T=c.T,
                              start_id=c.start_id, eos_id=c.eos_id,
                              tile_mode=TILE_MODES[c.tile_mode], mask_mode=MASK_MODES[c.mask_mode],
                              kv_from_normed=1 if c.kv_from_normed else 0,
                              precision=_lib.PRECISION_FP32,
                              length_penalty_weight=float(c.length_penalty_weight))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.icap_tengine_create(C.byref(cc), C.byref(w), C.byref(handle)))
        self._handle = handle

    @staticmethod
    def _expect(name, t, shape):
        if tuple(t.shape) != tuple(shape):
            raise ValueError("weight %s has shape %s, expected %s" % (name, tuple(t.shape), shape))

    def close(self):
        if self._handle is not None:
            _lib.load().icap_tengine_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(_lib.load().icap_tengine_workspace_bytes(self._handle))

    def decode(self, img, out=None, scores=None):
        """img [B,L,F] float32 -> (tokens [B,T] int32, scores [B])."""
        c = self.config
        if tuple(img.shape) != (c.B, c.L, c.F):
            raise ValueError("img must have shape %s" % ((c.B, c.L, c.F),))
        if out is None:
            out = torch.empty(c.B, c.T, dtype=torch.int32, device=self.device)
        if scores is None:
            scores = torch.empty(c.B, dtype=torch.float32, device=self.device)
        for name, t in (("img", img), ("out", out), ("scores", scores)):
            if t.is_cuda and t.device != self.device:
                raise ValueError("%s lives on %s, the engine on %s" % (name, t.device, self.device))
        ops._call("icap_tdecode", self._handle, ops._chk(img, torch.float32, "img"),
                  ops._chk(out, torch.int32, "out"), ops._chk(scores, torch.float32, "scores"),
                  ops.current_stream())
        return out, scores

    def trace(self):
        c = self.config
        shape = (c.T, c.B, c.W)
        out = {"ids": torch.empty(shape, dtype=torch.int32, device=self.device),
               "parents": torch.empty(shape, dtype=torch.int32, device=self.device),
               "scores": torch.empty(shape, dtype=torch.float32, device=self.device),
               "log_probs": torch.empty(shape, dtype=torch.float32, device=self.device)}
        with torch.cuda.device(self.device):
            _lib.call("icap_tengine_trace", self._handle, out["ids"].data_ptr(),
                      out["parents"].data_ptr(), out["scores"].data_ptr(), out["log_probs"].data_ptr(),
                      torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def last_output(self):
        """The stack's normalised output of the newest position at the last step [N, hidden]."""
        c = self.config
        t = torch.empty(c.N, c.hidden, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call("icap_tengine_last_output", self._handle, t.data_ptr(),
                      torch.cuda.current_stream(self.device).cuda_stream)
        return t

    def set_timing(self, kernels):
        if kernels is True:
            mask = (1 << len(_lib.TK_NAMES)) - 1
        elif not kernels:
            mask = 0
        else:
            mask = 0
            for k in kernels:
                mask |= 1 << _lib.TK_NAMES.index(k)
        _lib.call("icap_tengine_set_timing", self._handle, mask)

    def kernel_times(self):
        n = len(_lib.TK_NAMES)
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        _lib.call("icap_tengine_kernel_times", self._handle, ms, cnt)
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.TK_NAMES)}
