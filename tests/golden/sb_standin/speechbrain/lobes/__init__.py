from . import models  # noqa
