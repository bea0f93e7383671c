class Decoder:
    """Decoder interface (reference decoder/decoder.py:3-20)."""

    def __init__(self, vocab_size, data_type, name="decoder"):
        self.name = name
        self.vocab_size = vocab_size
        self.data_type = data_type

    def call(self, inputs, **kwargs):
        pass

    def __call__(self, inputs, **kwargs):
        return self.call(inputs, **kwargs)

    def reset_state(self, target, feat_d, **kwargs):
        pass

    def copy(self):
        pass

    def create_variables(self):
        pass
