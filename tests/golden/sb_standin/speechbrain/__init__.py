"""Stand-in for the parts of un-vendored SpeechBrain v1.0 that the reference overlay imports.

TEST INFRASTRUCTURE, used ONLY by tests/golden/make_golden.py inside the dev container
(where /root/reference exists). Written from the documented upstream semantics
(SURVEY.md §2.1); it is NOT reference source.  Package __path__s are extended so that the
UNMODIFIED reference files under /root/reference/speechbrain are imported in place.
"""
import os

_REF = os.environ.get("SMX_REFERENCE_ROOT", "/root/reference")
__path__.append(os.path.join(_REF, "speechbrain"))

from . import nnet, lobes, dataio, utils  # noqa: E402,F401
