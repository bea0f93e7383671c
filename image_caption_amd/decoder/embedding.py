from ..layers import Embedding


class EmbeddingLayer:
    """Token embedding lookup (reference decoder/embedding.py:3-10)."""

    def __init__(self, vocab_size, embedding_dim, device=None):
        self.embedding = Embedding(vocab_size, embedding_dim, device=device)

    @property
    def table(self):
        if not self.embedding.built:
            self.embedding.build()
        return self.embedding.table

    def call(self, inputs, **kwargs):
        return self.embedding(inputs)

    __call__ = call
