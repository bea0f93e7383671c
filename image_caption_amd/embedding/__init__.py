from .position_embedding import RelativePositionEmbedding  # noqa: F401
