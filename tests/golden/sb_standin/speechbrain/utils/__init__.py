from . import dynamic_chunk_training, checkpoints  # noqa
