"""KV caches for autoregressive decoding (interface of the reference's `models/kv_cache.py:3-69`).

MI355X layout: one (max_batch, max_seq, d) fp32 slab per layer for K and for V, written in place by the
projection GEMM of the current token (row stride = max_seq*d, no copy); the attention kernel reads the
first `pos+1` rows.  Unlike the reference (which stores the pre-`in_proj` rows and re-projects the
whole cache every step) the slabs hold the *projected* keys/values.  `VCache` holds the per-layer MSDA
value projection of the image memory, computed once per episode (the reference allocates it and never
reads it)."""
import torch
from torch import nn


class KVCache(nn.Module):
    def __init__(self, max_batch_size, max_seq_length, model_dim, dtype=torch.float32):
        super().__init__()
        shape = (max_batch_size, max_seq_length, model_dim)
        self.register_buffer("k_cache", torch.zeros(shape, dtype=dtype), persistent=False)
        self.register_buffer("v_cache", torch.zeros(shape, dtype=dtype), persistent=False)

    def update(self, input_pos, k_val, v_val):
        """Reference-compatible host API: write rows `input_pos`, return the valid prefixes."""
        index = int(input_pos[0]) + 1
        self.k_cache[:, input_pos, ...] = k_val
        self.v_cache[:, input_pos, ...] = v_val
        return self.k_cache[:, :index], self.v_cache[:, :index]


class VCache(nn.Module):
    def __init__(self, max_batch_size, max_seq_length, num_heads, head_dim, dtype=torch.float32):
        super().__init__()
        self.register_buffer("v_cache", torch.zeros((max_batch_size, max_seq_length, num_heads, head_dim), dtype=dtype),
                             persistent=False)

    def update(self, v_val):
        self.v_cache = v_val

    def get(self):
        return self.v_cache
