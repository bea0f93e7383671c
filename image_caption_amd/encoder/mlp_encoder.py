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
        self.fc = Dense(embedding_dim, name="{}/fc".format(self.name), device=device)
        self.activation = activation
        self.dropout = dropout

    def _is_relu(self):
        a = self.activation
        return a == "relu" or getattr(a, "__name__", "") in ("relu", "relu_")

    def call(self, inputs, **kwargs):
        """`activation`: "relu" or a relu function (the reference default, `:8`) is fused into
        the GEMM's epilogue; "linear" is the plain Dense; any other callable is applied to the
        Dense output as the reference does (`:17-18`).  None is not callable: TypeError at
        the call, as in the reference."""
        training = kwargs["training"]          # KeyError when omitted, like the reference
        if self._is_relu():
            x = self.fc(inputs, act=ops.ACT_RELU)
        elif self.activation == "linear":
            x = self.fc(inputs)
        elif callable(self.activation):
            x = self.activation(self.fc(inputs))
        else:
            raise TypeError("MlpEncoder: activation %r is not callable" % (self.activation,))
        if training and self.dropout > 0.0:
            raise NotImplementedError("MlpEncoder: dropout in training mode is outside the "
                                      "inference engine")
        return x
