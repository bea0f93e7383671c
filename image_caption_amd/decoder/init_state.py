from .. import ops
from ..layers import Dense


class FeatureProjState:
    """Initial recurrent state from pooled features (reference decoder/init_state.py:4-36):
    Dense(mean over L), or with a mask Dense(masked SUM over L) - the reference never
    divides by the mask count."""

    def __init__(self, units, name, device=None):
        self.fc = Dense(units, name=name, device=device)

    def __call__(self, features, mask=None, pooled=None):
        if pooled is None:
            pooled = ops.pool_features(features.contiguous(), mask)
        return self.fc(pooled)
