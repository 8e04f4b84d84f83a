"""Host utilities mirrored from the reference's `util/misc.py` interface (only what the CAPE hot
path touches): NestedTensor (:278-300), nested_tensor_from_tensor_list (:264-277), inverse_sigmoid
(:436-440), SmoothedValue / MetricLogger (:44-236) and the torch.distributed helpers
(:59-70, :128-153, :314-339).  Pure host glue -- no kernel work happens here."""
import datetime
import time
from collections import defaultdict, deque
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device, non_blocking=False):
        m = self.mask.to(device, non_blocking=non_blocking) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), m)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


_ZERO_MASKS = {}


def cached_zero_mask(n, h, w, device, dtype=torch.bool):
    """All-False (n, h, w) padding mask, one persistent tensor per geometry: a batch tensor of equally sized images has no
    padding at any pyramid level, so the per-step mask resampling of the reference (backbone.py:94-97, roomformer_v2.py:
    192-199: F.interpolate of an all-False mask) is a constant.  Read-only by convention."""
    key = (int(n), int(h), int(w), str(device), dtype)
    m = _ZERO_MASKS.get(key)
    if m is None:
        m = _ZERO_MASKS[key] = torch.zeros(n, h, w, dtype=dtype, device=device)
        m._cape_all_false = True                 # lets position-embedding code treat results derived from it as constants
    return m


def nested_tensor_from_tensor_list(tensor_list):
    """Pad a list (or batch tensor) of CHW images to a common size; mask is True on padded pixels."""
    if isinstance(tensor_list, Tensor) and tensor_list.ndim == 4:
        b, c, h, w = tensor_list.shape
        nt = NestedTensor(tensor_list, cached_zero_mask(b, h, w, tensor_list.device))
        nt.no_padding = True
        return nt
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    sizes = [list(img.shape) for img in tensor_list]
    c = sizes[0][0]
    h = max(s[1] for s in sizes)
    w = max(s[2] for s in sizes)
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    tensor = torch.zeros((len(tensor_list), c, h, w), dtype=dtype, device=device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[:, : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors (sorted keys, one stacked tensor); identity at world size 1."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([torch.as_tensor(input_dict[k], dtype=torch.float32).reshape(()).to(
            next(v.device for v in input_dict.values() if isinstance(v, Tensor))) for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}


class SmoothedValue(object):
    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count = int(t[0])
        self.total = t[1]

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        i = 0
        header = header or ""
        start = time.time()
        for obj in iterable:
            yield obj
            if i % print_freq == 0 or i == len(iterable) - 1:
                el = time.time() - start
                print(f"{header} [{i}/{len(iterable)}] {self} elapsed {datetime.timedelta(seconds=int(el))}")
            i += 1
