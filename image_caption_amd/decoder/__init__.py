from .decoder import Decoder  # noqa: F401
from .embedding import EmbeddingLayer  # noqa: F401
from .init_state import FeatureProjState  # noqa: F401
from .lstm_recurrent_decoder import LstmDecoder  # noqa: F401
from .recurrent_decoder import RecurrentDecoder  # noqa: F401
from .transform_decoder import TransformDecoder  # noqa: F401
