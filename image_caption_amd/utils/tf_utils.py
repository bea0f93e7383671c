"""The one helper of the reference's utils/tf_utils.py the decode path touches."""
import inspect


def get_kwargs():
    """Constructor arguments of the calling frame, minus self
    (reference utils/tf_utils.py:496-503; used by LstmDecoder.copy)."""
    frame = inspect.currentframe().f_back
    keys, _, _, values = inspect.getargvalues(frame)
    return {k: values[k] for k in keys if k != "self"}
