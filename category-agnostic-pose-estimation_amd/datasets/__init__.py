"""Input side of the CAPE path: tokenisation, episodic sampling / collation, the MP-100 file loader (COCO-style annotations
read by a pure-Python reader, transforms as plans that run on the host or on the GPU) and seeded synthetic episodes."""
from .discrete_tokenizer import DiscreteTokenizer, DiscreteTokenizerV2
from .episodic_sampler import EpisodicDataset, EpisodicSampler, build_episodic_dataloader, episodic_collate_fn
from .keypoint_tokenization import tokenize_keypoints
from .token_types import TokenType
from .mp100_cape import ImageNotFoundError, MP100CAPE, build_mp100_cape
