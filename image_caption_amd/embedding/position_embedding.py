from .. import ops


class RelativePositionEmbedding:
    """Sinusoid position encoding (reference embedding/position_embedding.py:27-102):
    call(inputs, length=None) -> [length, hidden_size] float32, [sin | cos] halves with
    timescales geometric between min_timescale and max_timescale."""

    def __init__(self, hidden_size, min_timescale=1.0, max_timescale=1.0e4, **kwargs):
        if (min_timescale, max_timescale) != (1.0, 1.0e4):
            raise NotImplementedError("RelativePositionEmbedding: only the default timescales "
                                      "(1 .. 1e4) are built")
        self._hidden_size = hidden_size
        self._min_timescale = min_timescale
        self._max_timescale = max_timescale

    def get_config(self):
        return {"hidden_size": self._hidden_size, "min_timescale": self._min_timescale,
                "max_timescale": self._max_timescale}

    def call(self, inputs, length=None):
        if inputs is None and length is None:
            raise ValueError("If inputs is None, `length` must be set in RelativePositionEmbedding().")
        device = None
        if inputs is not None:
            if length is not None and length != inputs.shape[1]:
                raise ValueError("If inputs is not None, `length` must equal to input_shape[1].")
            length = inputs.shape[1]
            device = inputs.device
        from ..layers import default_device
        return ops.position_table(length, self._hidden_size, device or default_device())

    __call__ = call
