from .. import ops
from ..layers import Dense
from .encoder import Encoder


class MlpEncoder(Encoder):
    """Dense(F -> embedding_dim) + ReLU + dropout over CNN features
    (reference encoder/mlp_encoder.py:6-21).  The GEMM, bias and ReLU are one HIP
    kernel.  `training` is a mandatory keyword as in the reference (`:19`)."""

    def __init__(self, embedding_dim, activation="relu", dropout=0.0, name="mlp_encoder",
                 device=None):
        super().__init__(name=name)
        if activation not in ("relu", None) and getattr(activation, "__name__", "") != "relu":
            raise NotImplementedError("MlpEncoder: only ReLU (the reference default) is built")
        self.fc = Dense(embedding_dim, name="{}/fc".format(self.name), device=device)
        self.activation = activation
        self.dropout = dropout

    def call(self, inputs, **kwargs):
        training = kwargs["training"]          # KeyError when omitted, like the reference
        if training and self.dropout > 0.0:
            raise NotImplementedError("MlpEncoder: dropout in training mode is outside the "
                                      "inference engine")
        return self.fc(inputs, act=ops.ACT_RELU)
