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


def assert_close(got, ref, rtol=1e-5, what="", allow_inf=False):
    """|got - ref| <= rtol * max(|ref|, scale) elementwise, scale = RMS of the finite part of
    ref.  The fp32 bar of the path: 1e-5 relative.  NaN anywhere fails; infinities fail unless
    allow_inf, and then they must sit at the same places with the same sign on both sides."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert not np.isnan(got).any(), "%s: %d NaN in the result" % (what, np.isnan(got).sum())
    assert not np.isnan(ref).any(), "%s: %d NaN in the expected values" % (what, np.isnan(ref).sum())
    inf_g, inf_r = np.isinf(got), np.isinf(ref)
    if not allow_inf:
        assert not inf_g.any() and not inf_r.any(), "%s: infinite values (%d got, %d expected)" % (
            what, inf_g.sum(), inf_r.sum())
    else:
        assert (inf_g == inf_r).all() and (np.sign(got[inf_g]) == np.sign(ref[inf_r])).all(), \
            "%s: infinities differ" % what
    fin = ~inf_r
    g, r = got[fin], ref[fin]
    scale = float(np.sqrt(np.mean(r ** 2))) if r.size else 1.0
    tol = rtol * np.maximum(np.abs(r), scale if scale > 0 else 1.0)
    bad = ~(np.abs(g - r) <= tol)
    assert not bad.any(), "%s: %d/%d off, max abs err %.3e (scale %.3e)" % (
        what, bad.sum(), bad.size, np.abs(g - r).max(), scale)


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


# ---- whole-sequence parity: the oracle follows the engine through proven near-ties ----------
#
# Both sides are fp32 with different summation orders, so a top-k decision between candidates
# whose scores are closer than fp32 noise may legitimately come out differently.  Instead of
# excusing an image after its first divergence, the oracle is run in lockstep with the engine's
# per-step records: wherever the engine's successors differ from the oracle's own top-k, every
# engine pick must be a near-tie AT ITS RANK - its score, computed by the oracle, within `bar`
# (relative) of the oracle's pick at the same rank, and the picks must be distinct candidates.
# Then the oracle adopts the engine's picks and goes on, so every later step of every image is
# still checked (ids, parents, log-probs, scores).  Anything else is a failure.

TIE_BAR = 2e-6          # about 30 fp32 ulps of a score


def following_chooser(tr_ids, tr_parents, V, bar=TIE_BAR):
    """choose_successors_fn for the oracle loop + the report it fills: `ties` = (step, image,
    rank, relative gap) per adopted pick, `failures` = the same for picks that are no near-tie,
    `min_margin` = the smallest relative gap between neighbours among the oracle's best k+1
    candidates over all steps and images (how far the closest decision was from a tie)."""
    report = {"ties": [], "failures": [], "min_margin": np.inf}
    step = [0]

    def choose(scores_flat, k):
        t = step[0]
        step[0] += 1
        vals1, idx1 = co.choose_top_k(scores_flat, min(k + 1, scores_flat.shape[-1]))
        s = vals1.astype(np.float64)
        ok = s > -1e30
        with np.errstate(invalid="ignore", over="ignore"):
            gaps = np.abs(np.diff(s, axis=1)) / np.maximum(np.abs(s[:, :-1]), 1e-30)
        gaps = np.where(ok[:, 1:] & ok[:, :-1], gaps, np.inf)
        if gaps.size:
            report["min_margin"] = min(report["min_margin"], float(gaps.min()))
        vals, idx = vals1[:, :k].copy(), idx1[:, :k].copy()
        idx_g = tr_parents[t].astype(np.int64) * V + tr_ids[t].astype(np.int64)
        for b in np.nonzero((idx != idx_g).any(axis=1))[0]:
            in_range = (idx_g[b] >= 0).all() and (idx_g[b] < scores_flat.shape[1]).all()
            sg = scores_flat[b, idx_g[b]] if in_range else np.full(k, np.nan, np.float32)
            with np.errstate(invalid="ignore", over="ignore"):
                gap = np.where(sg == vals[b], 0.0,
                               np.abs(sg.astype(np.float64) - vals[b]) /
                               np.maximum(np.abs(vals[b].astype(np.float64)), 1e-30))
            ok = in_range and len(set(idx_g[b].tolist())) == k and bool((gap <= bar).all())
            rows = [(t, int(b), int(r), float(gap[r])) for r in range(k) if idx[b, r] != idx_g[b, r]]
            if ok:
                report["ties"] += rows
                idx[b] = idx_g[b]
                vals[b] = sg
            else:
                report["failures"] += rows
        return vals, idx.astype(np.int32)

    return choose, report


def check_decode_following(p, img, W, T, start_id, eos_id, lpw, tile_mode, tok_g, tr_g, cell="lstm",
                           bar=TIE_BAR, rtol=1e-5, oracle_fn=None, return_report=False):
    """Runs the oracle loop following the engine's records (tr_g: ids, parents, scores,
    log_probs as numpy [T,B,W]; tok_g [B,T]) and asserts the whole-sequence parity bar.
    Returns (images that needed at least one near-tie adoption, the adopted ties)."""
    V = p["fc2_kernel"].shape[1] if "fc2_kernel" in p else p["t_fc2_kernel"].shape[1]
    choose, report = following_chooser(tr_g["ids"], tr_g["parents"], V, bar)
    if oracle_fn is None:
        tok_o, trace_o = co.predict_image_caption_sequence(
            p, img, W, T, start_id, eos_id, lpw, tile_mode, return_trace=True, hoist_w1=True,
            cell=cell, choose_successors_fn=choose)
    else:
        tok_o, trace_o = oracle_fn(choose)
    assert not report["failures"], "engine picks that are no near-tie of the oracle's (step, " \
        "image, rank, relative gap): %s" % report["failures"][:10]
    ids_o = np.stack([st["out"].predicted_ids for st in trace_o])
    par_o = np.stack([st["out"].beam_parent_ids for st in trace_o])
    np.testing.assert_array_equal(tr_g["ids"], ids_o)
    np.testing.assert_array_equal(tr_g["parents"], par_o)
    lp_o = np.stack([st["state"].log_probs for st in trace_o])
    sc_o = np.stack([st["out"].scores for st in trace_o])
    # (finished beams' masked candidates are about -3.4e38: compare those as "both below -1e30")
    with np.errstate(over="ignore", invalid="ignore"):
        assert_close(np.maximum(tr_g["log_probs"], -1e30), np.maximum(lp_o, -1e30), rtol=rtol,
                     what="log_probs")
        assert_close(np.maximum(tr_g["scores"], -1e30), np.maximum(sc_o, -1e30), rtol=rtol,
                     what="scores")
    # the returned caption: the oracle's back-tracking over the (followed) records, or - if the
    # last step's best beam is itself a near-tie - the path of a beam within the bar of the best
    B = tok_g.shape[0]
    out, state = trace_o[-1]["out"], trace_o[-1]["state"]
    final_ties = []
    for b in np.nonzero((tok_g != tok_o).any(axis=1))[0]:
        s = out.scores[b].astype(np.float64)
        if state.finished[b].any():
            s = s + (1.0 - state.finished[b]) * -1e9
        best = s.max()
        match = False
        for k in np.nonzero(np.abs(s - best) <= bar * abs(best))[0]:
            beam, path = int(k), []
            for t in range(T - 1, -1, -1):
                path.append(int(ids_o[t, b, beam]))
                beam = int(par_o[t, b, beam])
            match = match or path[::-1] == tok_g[b].tolist()
        assert match, "image %d: returned caption is not the best beam's path (nor a tied one)" % b
        final_ties.append((T, int(b), -1, 0.0))
    tied_images = sorted({b for _, b, _, _ in report["ties"] + final_ties})
    if return_report:
        return len(tied_images), report["ties"] + final_ties, report
    return len(tied_images), report["ties"] + final_ties
