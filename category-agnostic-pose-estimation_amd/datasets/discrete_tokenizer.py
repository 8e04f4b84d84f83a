"""Vocabulary of the discretised keypoint sequence (interface of the reference's
`datasets/discrete_tokenizer.py:3-125`): ids [0, num_bins^2) are grid cells x*num_bins+y, followed by
BOS, EOS, SEP, PAD (and CLS when add_cls)."""
import numpy as np
import torch


class DiscreteTokenizer(object):
    def __init__(self, num_bins, seq_len, add_cls=False):
        self.num_bins = num_bins
        self.seq_len = seq_len
        self.add_cls = add_cls
        grid = num_bins * num_bins
        self.bos, self.eos, self.sep, self.pad = grid, grid + 1, grid + 2, grid + 3
        if add_cls:
            self.cls = grid + 4
        self.vocab_size = grid + (5 if add_cls else 4)

    def __len__(self):
        return self.vocab_size

    def _pack(self, seq, add_bos, add_eos, skip_rather_than_stop):
        out = [self.bos] if add_bos else []
        kept = []
        extra = 2 if self.add_cls else 1
        for i, sub in enumerate(seq):
            if len(out) + len(sub) + extra <= self.seq_len:
                out.extend(sub)
                kept.append(i)
            elif skip_rather_than_stop:
                continue
            else:
                break
            if self.add_cls:
                out.append(self.cls)
            out.append(self.sep)
        if out and out[-1] == self.sep:
            out.pop()
        out.extend([self.pad] * (self.seq_len - len(out)))
        if add_eos:
            out[-1] = self.eos
        return out, kept

    def __call__(self, seq, add_bos, add_eos, dtype):
        return torch.tensor(self._pack(seq, add_bos, add_eos, False)[0], dtype=dtype)

    def _padding(self, seq, pad_value, dtype):
        seq = list(seq)
        if self.seq_len > len(seq):
            seq.extend([pad_value] * (self.seq_len - len(seq)))
        return torch.tensor(np.array(seq), dtype=dtype)


class DiscreteTokenizerV2(DiscreteTokenizer):
    """Variant that skips (instead of stopping at) a polygon that does not fit, and can report which were kept."""

    def __call__(self, seq, add_bos, add_eos, dtype, return_indices=False):
        out, kept = self._pack(seq, add_bos, add_eos, True)
        t = torch.tensor(out, dtype=dtype)
        return (t, kept) if return_indices else t
