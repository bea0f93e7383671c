from .. import ops
from ..layers import Dense


class FeedForwardNetwork:
    """Position-wise feed-forward network (reference decoder/ffn_layer.py:24-75):
    Dense(filter_size, relu) -> dropout (identity at inference) -> Dense(hidden_size)."""

    def __init__(self, hidden_size, filter_size, relu_dropout, device=None):
        self.hidden_size = hidden_size
        self.filter_size = filter_size
        self.relu_dropout = relu_dropout
        self.filter_dense_layer = Dense(filter_size, name="filter_layer", device=device)
        self.output_dense_layer = Dense(hidden_size, name="output_layer", device=device)

    def get_config(self):
        return {"hidden_size": self.hidden_size, "filter_size": self.filter_size,
                "relu_dropout": self.relu_dropout}

    def call(self, x, training):
        if training and self.relu_dropout > 0.0:
            raise NotImplementedError("FeedForwardNetwork: dropout in training mode is outside the "
                                      "inference engine")
        output = self.filter_dense_layer(x.contiguous(), act=ops.ACT_RELU)
        return self.output_dense_layer(output)

    __call__ = call
