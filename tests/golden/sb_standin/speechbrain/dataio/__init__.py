from . import dataio  # noqa
