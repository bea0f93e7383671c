from .encoder import Encoder  # noqa: F401
from .mlp_encoder import MlpEncoder  # noqa: F401
