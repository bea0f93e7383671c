class Encoder:
    """Encoder interface (reference encoder/encoder.py:4-11)."""

    def __init__(self, name="encoder"):
        self.name = name

    def call(self, inputs, **kwargs):
        pass

    def __call__(self, inputs, **kwargs):
        return self.call(inputs, **kwargs)
