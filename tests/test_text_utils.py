"""Detokenisation and sentence BLEU of the reference's dev loop (recurrent_example.py:766-783).
nltk is not installed here; the BLEU restatement is pinned by the worked example of nltk's own
sentence_bleu documentation (three references, hypothesis 1: 0.5045666840058485)."""
import numpy as np
import pytest

from image_caption_amd.utils.text import (brevity_penalty, closest_ref_length, corpus_mean_bleu,
                                          detokenize, detokenize_batch, modified_precision,
                                          sentence_bleu)

H1 = "It is a guide to action which ensures that the military always obeys the commands of the party".split()
R1 = "It is a guide to action that ensures that the military will forever heed Party commands".split()
R2 = ("It is the guiding principle which guarantees the military forces always being under the "
      "command of the Party").split()
R3 = "It is the practical guide for the army always to heed the directions of the party".split()


def test_sentence_bleu_known_answers():
    assert sentence_bleu([R1, R2, R3], H1) == pytest.approx(0.5045666840058485, rel=1e-12)
    assert sentence_bleu([R1], R1) == pytest.approx(1.0)
    assert sentence_bleu([R1], ["nothing", "in", "common", "here"]) == 0
    # clipped precision example of the BLEU paper: "the" x7 against two references
    hyp = ["the"] * 7
    refs = ["the cat is on the mat".split(), "there is a cat on the mat".split()]
    p1 = modified_precision(refs, hyp, 1)
    assert (p1.numerator, p1.denominator) == (2, 7)


def test_brevity_penalty_and_closest_length():
    assert closest_ref_length([[0] * 13, [0] * 2], 12) == 13
    assert closest_ref_length([[0] * 13, [0] * 11], 12) == 11       # tie -> the shorter one
    assert brevity_penalty(12, 12) == 1.0
    assert brevity_penalty(12, 13) == 1.0
    assert brevity_penalty(12, 6) == pytest.approx(np.exp(1 - 2.0))
    assert brevity_penalty(5, 0) == 0.0


def test_detokenize_drops_control_tokens():
    index_word = {0: "<pad>", 1: "<unk>", 3: "<s>", 4: "s>", 5: "a", 6: "dog", 7: "runs"}
    assert detokenize([3, 5, 6, 7, 4, 0, 0], index_word) == ["a", "dog", "runs"]
    assert detokenize_batch(np.array([[3, 5, 4], [3, 1, 6]]), index_word) == [["a"], ["<unk>", "dog"]]
    with pytest.raises(KeyError):
        detokenize([99], index_word)
    ref = [[3, 5, 6, 7, 5, 6, 4, 0]]
    assert corpus_mean_bleu(ref, ref, index_word) == pytest.approx(1.0)
    assert corpus_mean_bleu([], [], index_word) == 0.0


def test_tokenizer_table_follows_the_keras_rules():
    """fit_tokenizer (data/data_util.py:56-66): ids by descending count with ties in order of
    appearance, '<unk>' = 1, '<pad>' = 0, the markers '<s>' / '</s>' become '<s>' and '<', 's>'
    under the reference's filter string, ids >= top_k map to '<unk>'.  Parity unpinned
    (TensorFlow absent): checked against the documented behaviour of the Keras class."""
    from image_caption_amd.utils.text import fit_tokenizer, pad_sequences_post
    caps = ["<s> A dog runs. </s>", "<s> a DOG sits, a cat sleeps </s>", "<s> a bird </s>"]
    tok = fit_tokenizer(caps, top_k=8)
    assert tok.index_word[0] == "<pad>" and tok.index_word[1] == "<unk>"
    # counts: a 4; <s>, <, s> 3 each (that order of appearance); dog 2; then runs, sits, cat, ...
    assert [tok.index_word[i] for i in range(2, 10)] == ["a", "<s>", "<", "s>", "dog", "runs", "sits", "cat"]
    assert tok.word_index["s>"] == 5 and tok.word_index["<s>"] == 3
    seqs = tok.texts_to_sequences(["<s> a cat runs </s>", "unseen word"])
    assert seqs == [[3, 2, 1, 7, 4, 5], [1, 1]]          # cat (9) >= top_k -> <unk>; runs (7) kept
    vec = pad_sequences_post(seqs)
    assert vec.shape == (2, 6) and vec[1].tolist() == [1, 1, 0, 0, 0, 0]
