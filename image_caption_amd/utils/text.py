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
