"""`chainer.function_node.FunctionNode` stand-in: apply = unwrap -> forward -> wrap, recorded on the tape with
the node's own hand-written `backward(target_input_indexes, grad_outputs)` as the vector-Jacobian product
(that is how Chainer drives DiffLqr.backward / MPCstep.backward)."""
import numpy as np

from .variable import Variable, _raw, make_multi, recording


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
        if not recording() or not any(isinstance(x, Variable) for x in inputs):
            return tuple(Variable(o) for o in outs)
        n_in = len(inputs)

        def vjp(gys):
            gys = tuple(Variable(np.zeros_like(o)) if g is None else g for g, o in zip(gys, outs))
            gxs = self.backward(tuple(range(n_in)), gys)
            res = []
            for gx in gxs:
                if gx is None or (isinstance(gx, Variable) and gx.array is None):
                    res.append(None)
                else:
                    res.append(gx if isinstance(gx, Variable) else Variable(gx))
            return res

        return tuple(make_multi(outs, list(inputs), vjp))
