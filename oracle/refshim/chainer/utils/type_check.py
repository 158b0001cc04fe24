"""`chainer.utils.type_check` stand-in: checks are no-ops."""


def expect(*args, **kwargs):
    return None
