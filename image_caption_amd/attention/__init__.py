from .attention import BahdanauAttention  # noqa: F401
