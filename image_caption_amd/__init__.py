"""MI355X-native caption-decoding engine with the module interfaces of
king821221/image-caption's show-attend-tell inference path.

Sub-packages mirror the reference's module paths (`encoder.mlp_encoder`,
`decoder.lstm_recurrent_decoder`, `decoder.embedding`, `attention.attention`,
`beam_search.beam_search`); every tensor is a torch tensor on the ROCm device and
every operator runs as a hand-written HIP kernel behind the C ABI of libicap.so."""
from . import _lib
from .engine import CaptionEngine, sequence_beam_search  # noqa: F401

__all__ = ["CaptionEngine", "sequence_beam_search", "_lib"]
