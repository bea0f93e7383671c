from .beam_search import *  # noqa: F401,F403
