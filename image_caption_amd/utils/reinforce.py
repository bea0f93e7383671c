"""The decoder adapter the reference's sequence-level trainers hand to their samplers and
searchers (reference utils/reinforce.py:75-105); only the inference-side surface is built."""


class RnnDecoderNetwork(object):
    """Wraps a recurrent decoder so that a call returns (predictions, next_state) only."""

    def __init__(self, rnn_decoder):
        self.rnn_decoder = rnn_decoder

    def __call__(self, inputs, **kwargs):
        predictions, state_next, _, _ = self.rnn_decoder(inputs, **kwargs)
        return predictions, state_next

    def reset_state(self, sequence, feat_d, **kwargs):
        return self.rnn_decoder.reset_state(sequence, feat_d, **kwargs)

    def create_variables(self):
        return self.rnn_decoder.create_variables()

    def clone(self):
        return type(self)(self.rnn_decoder.copy())
