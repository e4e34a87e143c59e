from .attention import _Unsupported


class Embedding(_Unsupported):
    pass
