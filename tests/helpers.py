"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch

from oracle import caption_oracle as co


def dev():
    return torch.device("cuda", 0)


def to_dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def np_of(t):
    return t.detach().cpu().numpy()


def assert_close(got, ref, rtol=1e-5, what=""):
    """|got - ref| <= rtol * max(|ref|, scale) elementwise, scale = RMS of ref.
    The fp32 bar of the path: 1e-5 relative."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(np.sqrt(np.mean(ref ** 2))) if ref.size else 1.0
    tol = rtol * np.maximum(np.abs(ref), scale if scale > 0 else 1.0)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), "%s: %d/%d off, max abs err %.3e (scale %.3e)" % (
        what, bad.sum(), bad.size, np.abs(got - ref).max(), scale)


def state_to_dev(state):
    from image_caption_amd.beam_search import beam_search as bs
    return bs.BeamSearchState(to_dev(state.log_probs), to_dev(state.finished),
                              to_dev(state.lengths))


def round_bf16(x):
    """float32 array rounded to the nearest bfloat16 (ties to even), returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


def prefer_fused(eng):
    """Select the fused projection + beam step where the engine's shape is served (small
    batches default to the unfused kernels); returns whether it is on."""
    from image_caption_amd._lib import IcapError
    try:
        eng.set_fused_vocab(True)
    except IcapError:
        pass
    return eng.fused_vocab
