"""`chainer.function_node.FunctionNode` stand-in: apply = unwrap -> forward -> wrap."""
from .variable import Variable, _raw


class FunctionNode:
    def __init__(self):
        self._in_idx = ()
        self._out_idx = ()
        self._inputs = None
        self._outputs = None

    def retain_inputs(self, indexes):
        self._in_idx = tuple(indexes)

    def retain_outputs(self, indexes):
        self._out_idx = tuple(indexes)

    def get_retained_inputs(self):
        return tuple(None if self._inputs[i] is None else Variable(self._inputs[i])
                     for i in self._in_idx)

    def get_retained_outputs(self):
        return tuple(Variable(self._outputs[i]) for i in self._out_idx)

    def apply(self, inputs):
        raw = tuple(None if x is None else _raw(x) for x in inputs)
        self._inputs = raw
        outs = self.forward(raw)
        if not isinstance(outs, tuple):
            outs = (outs,)
        outs = tuple(_raw(o) for o in outs)
        self._outputs = outs
        return tuple(Variable(o) for o in outs)
