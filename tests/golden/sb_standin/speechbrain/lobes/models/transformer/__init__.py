import os
from .... import _REF
__path__.append(os.path.join(_REF, "speechbrain", "lobes", "models", "transformer"))
