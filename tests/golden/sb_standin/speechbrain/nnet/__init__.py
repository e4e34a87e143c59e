import os
from .. import _REF
__path__.append(os.path.join(_REF, "speechbrain", "nnet"))
from . import containers, linear, normalization, activations, attention, hypermixing, CNN, embedding  # noqa
