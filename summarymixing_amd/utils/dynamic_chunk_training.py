"""DynChunkTrainConfig with the fields the encoder reads (speechbrain.utils.dynamic_chunk_training)."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class DynChunkTrainConfig:
    chunk_size: int
    left_context_size: Optional[int] = None

    def is_infinite_left_context(self) -> bool:
        return self.left_context_size is None

    def left_context_size_frames(self) -> Optional[int]:
        return None if self.left_context_size is None else self.chunk_size * self.left_context_size
