"""Input contract of the CAPE hot path (tokenisation, episodic collation, synthetic episodes).  The
MP-100 file loaders of the reference (`datasets/mp100_cape.py` image I/O, pycocotools, albumentations)
are host I/O outside the hot path."""
from .discrete_tokenizer import DiscreteTokenizer, DiscreteTokenizerV2
from .episodic_sampler import episodic_collate_fn
from .keypoint_tokenization import tokenize_keypoints
from .token_types import TokenType
