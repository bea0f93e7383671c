"""Host-side evaluation helpers of the reference's dev loop (recurrent_example.py:766-777):
id -> word mapping with the control tokens dropped, and the sentence BLEU the loop reports
(nltk.translate.bleu_score.sentence_bleu with its defaults: 4-gram uniform weights, clipped
n-gram precision, brevity penalty against the closest reference length, no smoothing)."""
from __future__ import annotations

import math
import sys
from collections import Counter

CONTROL_WORDS = ('s>', '<s>', '<pad>')      # end, start, pad as the reference spells them


def detokenize(token_ids, index_word, drop=CONTROL_WORDS):
    """One row of token ids -> list of words, control tokens removed
    (recurrent_example.py:770-771).  Raises KeyError for an id outside the mapping, as the
    reference's dict lookup does."""
    words = [index_word[int(t)] for t in token_ids]
    return [w for w in words if w not in drop]


def detokenize_batch(tokens, index_word, drop=CONTROL_WORDS):
    """tokens [B,T] (tensor, array or nested lists) -> B word lists."""
    rows = tokens.tolist() if hasattr(tokens, "tolist") else tokens
    return [detokenize(r, index_word, drop) for r in rows]


def _ngrams(seq, n):
    return [tuple(seq[i:i + n]) for i in range(len(seq) - n + 1)]


def modified_precision(references, hypothesis, n):
    """Clipped n-gram precision as an unreduced numerator / denominator pair (numerator: hypothesis n-gram counts clipped by
    the maximum count in any reference; denominator: hypothesis n-grams, at least 1)."""
    counts = Counter(_ngrams(hypothesis, n)) if len(hypothesis) >= n else Counter()
    max_counts = {}
    for ref in references:
        ref_counts = Counter(_ngrams(ref, n)) if len(ref) >= n else Counter()
        for g in counts:
            max_counts[g] = max(max_counts.get(g, 0), ref_counts[g])
    clipped = {g: min(c, max_counts[g]) for g, c in counts.items()}
    return _RawFraction(sum(clipped.values()), max(1, sum(counts.values())))


class _RawFraction:
    """Numerator / denominator pair that is not reduced (the zero-numerator test and the
    logarithm only need the two integers)."""

    def __init__(self, numerator, denominator):
        self.numerator, self.denominator = numerator, denominator


def closest_ref_length(references, hyp_len):
    return min((len(r) for r in references), key=lambda rl: (abs(rl - hyp_len), rl))


def brevity_penalty(closest_ref_len, hyp_len):
    if hyp_len > closest_ref_len:
        return 1.0
    if hyp_len == 0:
        return 0.0
    return math.exp(1 - closest_ref_len / hyp_len)


def sentence_bleu(references, hypothesis, weights=(0.25, 0.25, 0.25, 0.25)):
    """BLEU of one hypothesis (list of words) against a list of references.  No 1-gram
    overlap gives 0; a higher order without overlap contributes log(smallest positive
    float), which makes the score vanishingly small, as the unsmoothed nltk score is."""
    p_n = [modified_precision(references, hypothesis, i) for i in range(1, len(weights) + 1)]
    if p_n[0].numerator == 0:
        return 0
    bp = brevity_penalty(closest_ref_length(references, len(hypothesis)), len(hypothesis))
    s = 0.0
    for w, p in zip(weights, p_n):
        value = p.numerator / p.denominator if p.numerator != 0 else sys.float_info.min
        s += w * math.log(value)
    return bp * math.exp(s)


def corpus_mean_bleu(reference_rows, hypothesis_rows, index_word):
    """Mean sentence BLEU of token-id rows against one reference row each, as the dev loop
    aggregates it (recurrent_example.py:766-783)."""
    scores = []
    for ref_ids, hyp_ids in zip(reference_rows, hypothesis_rows):
        real = detokenize(ref_ids, index_word)
        result = detokenize(hyp_ids, index_word)
        scores.append(sentence_bleu([real], result))
    return sum(scores) / len(scores) if scores else 0.0


# ---- the tokenizer of the data pipeline (data/data_util.py:56-72) ---------------------------

KERAS_FILTERS = '!"#$%&()*+.,-/:;=?@[\\]^_`{|}~'   # the reference's filter string (no '<', no '>')


class Tokenizer:
    """The id <-> word table that `fit_tokenizer` builds with `tf.keras.preprocessing.text.
    Tokenizer(num_words=top_k, oov_token="<unk>", filters=...)`, restated from the published
    behaviour of that class (TensorFlow is not installed here, so the table is checked against
    the class's documented rules, not against its output):
    texts are lower-cased, every filter character becomes a space, words are split on
    whitespace; words are ranked by count, descending, ties in order of first appearance; the
    out-of-vocabulary token takes index 1 and the ranked words follow from 2; `num_words` does
    not shrink the table, it only maps ids >= num_words to the oov id in `texts_to_sequences`;
    the reference then adds '<pad>' as id 0 (`data/data_util.py:64-65`).  With the reference's
    filter string the caption markers '<s>' and '</s>' come out as the words '<s>' and
    ('<', 's>'), which is why its end token is spelled 's>'."""

    def __init__(self, num_words=None, oov_token="<unk>", filters=KERAS_FILTERS, lower=True):
        self.num_words, self.oov_token, self.filters, self.lower = num_words, oov_token, filters, lower
        self.word_counts = {}
        self.word_index = {}
        self.index_word = {}

    def _words(self, text):
        if self.lower:
            text = text.lower()
        return text.translate({ord(c): " " for c in self.filters}).split()

    def fit_on_texts(self, texts):
        for t in texts:
            for w in self._words(t):
                self.word_counts[w] = self.word_counts.get(w, 0) + 1
        ranked = sorted(self.word_counts.items(), key=lambda kv: kv[1], reverse=True)   # stable
        vocab = ([self.oov_token] if self.oov_token is not None else []) + [w for w, _ in ranked]
        self.word_index = {w: i + 1 for i, w in enumerate(vocab)}
        self.index_word = {i: w for w, i in self.word_index.items()}

    def texts_to_sequences(self, texts):
        oov = self.word_index.get(self.oov_token)
        out = []
        for t in texts:
            seq = []
            for w in self._words(t):
                i = self.word_index.get(w)
                if i is not None and (self.num_words is None or i < self.num_words):
                    seq.append(i)
                elif oov is not None:
                    seq.append(oov)
            out.append(seq)
        return out


def fit_tokenizer(captions, top_k):
    """`fit_tokenizer(captions, top_k=...)`, data/data_util.py:56-66."""
    tok = Tokenizer(num_words=top_k)
    tok.fit_on_texts(captions)
    tok.word_index["<pad>"] = 0
    tok.index_word[0] = "<pad>"
    return tok


def pad_sequences_post(seqs, value=0):
    """`pad_sequences(seqs, padding='post')` (data/data_util.py:70-71): int32 [n, longest]."""
    import numpy as np
    width = max((len(s) for s in seqs), default=0)
    out = np.full((len(seqs), width), value, np.int32)
    for r, s in enumerate(seqs):
        out[r, :len(s)] = s
    return out
