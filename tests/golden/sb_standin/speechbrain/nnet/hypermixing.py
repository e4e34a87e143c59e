from .attention import _Unsupported


class HyperMixing(_Unsupported):
    pass
