"""The usable captioner end to end on the GPU: tokenizer table -> weights through the engine's
.npz container -> fused decode -> id -> word mapping -> the dev loop's sentence BLEU
(recurrent_example.py:359-371 weight source, data/data_util.py:56-72 tokenizer, :766-783
detokenisation and BLEU).  Tokens must equal the oracle's; the text utilities are host code."""
import numpy as np
import pytest
import torch

from oracle import caption_oracle as co
from tests.helpers import dev, np_of, to_dev

pytestmark = pytest.mark.gpu

CAPTIONS = ["<s> a man rides a wave on a surfboard </s>", "<s> a dog runs on the beach </s>",
            "<s> two people sit at a table with food </s>", "<s> a cat sleeps on a red couch </s>",
            "<s> a plane flies over the city at night </s>", "<s> a woman holds an umbrella in the rain </s>",
            "<s> a train waits at the station </s>", "<s> children play ball in a green park </s>"]


def test_npz_weights_decode_detokenize_bleu(tmp_path):
    from image_caption_amd import synthetic
    from image_caption_amd.engine import CaptionConfig, CaptionEngine
    from image_caption_amd.utils import text
    tok = text.fit_tokenizer(CAPTIONS, top_k=5000)
    start, eos = tok.word_index["<s>"], tok.word_index["s>"]
    V = (len(tok.index_word) + 3) // 4 * 4                      # the fused projection wants V % 4 == 0
    for i in range(len(tok.index_word), V):
        tok.index_word[i] = "<unk>"
    dims = dict(F=64, D=32, E=32, H=64, V=V)
    p = co.perturb_biases(co.make_peaked_params(2, eos_id=eos, **dims), 3)
    path = tmp_path / "captioner.npz"
    synthetic.save_params(str(path), p)
    loaded = synthetic.load_params(str(path))
    assert sorted(loaded) == sorted(p) and all(loaded[k].dtype == np.float32 for k in loaded)
    B, W, L, T = 8, 3, 6, 9
    img = co.make_images(11, B, L, dims["F"])
    ref = co.predict_image_caption_sequence(p, img, W, T, start, eos, 1.0, "repeat")
    cfg = CaptionConfig(B=B, W=W, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"], U=dims["H"],
                        V=V, T=T, start_id=start, eos_id=eos, tile_mode="repeat")
    eng = CaptionEngine(cfg, loaded, device=dev())
    tokens, _ = eng.decode(to_dev(img), check=True)
    got = np_of(tokens)
    eng.close()
    np.testing.assert_array_equal(got, ref)
    words = text.detokenize_batch(tokens, tok.index_word)        # straight from the device tensor
    assert words == text.detokenize_batch(ref, tok.index_word)
    assert all(w not in text.CONTROL_WORDS for row in words for w in row)
    assert len({tuple(r) for r in words}) > 1                     # the captions differ between images
    target = text.pad_sequences_post(tok.texts_to_sequences(CAPTIONS))
    bleu = text.corpus_mean_bleu(target, got, tok.index_word)
    assert 0.0 <= bleu <= 1.0
    assert text.corpus_mean_bleu(got, got, tok.index_word) == pytest.approx(
        np.mean([1.0 if len(r) >= 4 else 0.0 for r in words]))   # a sentence against itself
