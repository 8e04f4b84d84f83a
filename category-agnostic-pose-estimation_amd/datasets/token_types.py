"""Token classes of the keypoint sequence (reference `datasets/token_types.py`)."""
from enum import Enum


class TokenType(Enum):
    coord = 0
    sep = 1
    eos = 2
    cls = 3
