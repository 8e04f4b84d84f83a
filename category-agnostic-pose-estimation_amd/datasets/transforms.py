"""Image / keypoint transforms of the MP-100 loader, MI355X-first.

The reference pipes every crop through albumentations on the host (`datasets/mp100_cape.py:896-950`: train = Affine
(translate +-10 %, scale 0.85-1.15, rotate +-30 deg, p 0.7) -> HorizontalFlip(0.5) -> ColorJitter(0.3, 0.3, 0.3, 0.1, p 0.6) ->
OneOf(GaussNoise, GaussianBlur(3-7), MotionBlur(5), p 0.3) -> Resize(512); val / test = Resize(512)).  Here a transform is a
small *plan* drawn on the host -- the random numbers of those transforms, in the reference's order and spaces: the affine map and
the flip act on the crop before the resize, the colour ops and the blur / noise on the crop-sized image, the resize comes last --
and keypoints follow the plan's forward map on the host (a few numbers).  Pixels are produced either
  * on the GPU (`DeviceImagePipeline` -> `cape_augment_batch`, csrc/augment.hip): the DataLoader workers ship the raw uint8 crops
    of a batch as ONE pinned buffer plus the plans, two launches make the (N, 3, S, S) fp32 batch on the pipeline's stream while
    the previous batch trains -- the host cores only decode and crop; or
  * on the host (`apply_plan_host`, the same arithmetic in torch CPU ops: the CPU tests, `CAPE_HOST_AUGMENT=1`).
Both produce the same pixels for the same plan (tests/test_augment_gpu.py).  Known deviations from albumentations (its pixel
parity is unpinned: the library is not installed, and cv2's fixed-point uint8 arithmetic is not reproduced): float arithmetic
with a clip to [0, 1] after each colour op instead of uint8 rounding; the contrast jitter pivots on the mean grey of the warped
image (times the brightness factor when brightness precedes it) rather than of the running uint8 image; GaussNoise uses
var_limit (10, 50) on the 0-255 scale; the motion-blur kernel is a random line through the k x k window."""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

GAUSS_K = (3, 5, 7)


class TransformPlan:
    """`fwd` (3x3): source crop pixel index (x, y) -> augmented pixel index (affine about the centre, then flip), both at crop
    resolution (h, w); the output is the augmented image resized to out_size.  Keypoints: fwd, then * out_size / (w, h)."""

    def __init__(self, h, w, out_size, fwd=None, color=None, mode=0, noise_std=0.0, noise_seed=0, blur_kernel=None, flipped=False):
        self.h, self.w, self.out_size = int(h), int(w), int(out_size)
        self.fwd = np.eye(3) if fwd is None else np.asarray(fwd, dtype=np.float64).reshape(3, 3)
        self.inv = np.linalg.inv(self.fwd)                   # aug pixel index -> source pixel index (what the sampler needs)
        self.color = color                                   # None or (order (4 ints), brightness, contrast, saturation, hue)
        self.mode, self.noise_std, self.noise_seed = int(mode), float(noise_std), int(noise_seed)
        self.blur_kernel = None if blur_kernel is None else np.asarray(blur_kernel, dtype=np.float32)
        self.flipped = bool(flipped)

    def map_keypoints(self, kpts):
        """Keypoints are continuous coordinates (pixel i covers [i, i + 1)); `fwd` acts on pixel indices (centres at integers)."""
        k = np.asarray(kpts, dtype=np.float64).reshape(-1, 2)
        sx, sy = self.out_size / self.w, self.out_size / self.h
        if np.array_equal(self.fwd, np.eye(3)):              # Resize only: exactly the reference's k * size / (w, h)
            return [(float(x * sx), float(y * sy)) for x, y in k]
        out = (np.c_[k - 0.5, np.ones(len(k))] @ self.fwd.T)[:, :2] + 0.5
        return [(float(x * sx), float(y * sy)) for x, y in out]


def resize_plan(h, w, size=512):
    """albumentations.Resize(size, size): keypoints scale by size/w, size/h (`mp100_cape.py:942-944`)."""
    return TransformPlan(h, w, size)


def _motion_kernel(k, rng):
    """Random line through the k x k window (albumentations.MotionBlur draws a line between two random points), normalised."""
    ker = np.zeros((k, k), dtype=np.float32)
    while True:
        x0, y0, x1, y1 = (int(v) for v in rng.integers(0, k, 4))
        if (x0, y0) != (x1, y1):
            break
    n = max(abs(x1 - x0), abs(y1 - y0))
    for t in range(n + 1):
        ker[int(round(y0 + (y1 - y0) * t / n)), int(round(x0 + (x1 - x0) * t / n))] = 1.0
    return ker / ker.sum()


def _gauss_kernel(k):
    """cv2.getGaussianKernel with sigma = 0 (derived from the size, as albumentations' GaussianBlur with sigma_limit 0)."""
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = np.arange(k, dtype=np.float64) - (k - 1) / 2
    g = np.exp(-x * x / (2 * sigma * sigma))
    g /= g.sum()
    return np.outer(g, g).astype(np.float32)


def train_plan(h, w, rng, size=512):
    """Random plan with the reference's training distribution (`mp100_cape.py:896-940`), drawn from `rng` (numpy Generator)."""
    A = np.eye(3)
    if rng.random() < 0.7:                                   # A.Affine(translate_percent +-0.1, scale 0.85-1.15, rotate +-30, p 0.7)
        ang = math.radians(rng.uniform(-30, 30))
        sx, sy = rng.uniform(0.85, 1.15), rng.uniform(0.85, 1.15)          # (keep_ratio=False: one draw per axis)
        tx, ty = rng.uniform(-0.1, 0.1) * w, rng.uniform(-0.1, 0.1) * h
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        R = np.array([[math.cos(ang), -math.sin(ang), 0.0], [math.sin(ang), math.cos(ang), 0.0], [0.0, 0.0, 1.0]])
        Sc = np.diag([sx, sy, 1.0])
        T0, T1 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]]), np.array([[1, 0, cx + tx], [0, 1, cy + ty], [0, 0, 1.0]])
        A = T1 @ R @ Sc @ T0
    flipped = rng.random() < 0.5                             # A.HorizontalFlip(p 0.5)
    if flipped:
        A = np.array([[-1, 0, w - 1.0], [0, 1, 0], [0, 0, 1.0]]) @ A
    color = None
    if rng.random() < 0.6:                                   # A.ColorJitter(0.3, 0.3, 0.3, 0.1, p 0.6): the four ops in random order
        order = [int(v) for v in rng.permutation(4)]
        color = (order, rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3), rng.uniform(-0.1, 0.1))
    mode, noise_std, ker = 0, 0.0, None
    if rng.random() < 0.3:                                   # A.OneOf([GaussNoise, GaussianBlur(3-7), MotionBlur(5)], p 0.3)
        which = int(rng.integers(3))
        if which == 0:
            mode, noise_std = 1, math.sqrt(rng.uniform(10.0, 50.0)) / 255.0
        elif which == 1:
            mode, ker = 2, _gauss_kernel(GAUSS_K[int(rng.integers(3))])
        else:
            mode, ker = 2, _motion_kernel((3, 5)[int(rng.integers(2))], rng)
    return TransformPlan(h, w, size, A, color, mode, noise_std, int(rng.integers(1 << 31)), ker, flipped)


# ------------------------------------------------------------------------------------------------
# host implementation (same arithmetic as csrc/augment.hip, in torch CPU ops)
# ------------------------------------------------------------------------------------------------
def _rng_u32(seed, stream, idx):
    """cape_rng_u32(seed, step = 0, stream, idx) of csrc/common.h on numpy uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(1) + np.uint64(0xD1342543DE82EF95) * np.uint64(stream + 1)
        x = x ^ (idx.astype(np.uint64) * np.uint64(0xA0761D6478BD642F))
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


def _gauss_noise(seed, h, w):
    idx = np.arange(h * w, dtype=np.uint64)
    out = np.empty((3, h * w), dtype=np.float32)
    for ch in range(3):
        a, c = _rng_u32(seed, ch, 2 * idx), _rng_u32(seed, ch, 2 * idx + 1)
        u1 = (a.astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 4294967296.0)
        u2 = c.astype(np.float32) * np.float32(1.0 / 4294967296.0)
        out[ch] = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return torch.from_numpy(out.reshape(3, h, w))


def _grey(x):
    return 0.299 * x[0] + 0.587 * x[1] + 0.114 * x[2]


def _hue_shift(x, shift):
    r, g, b = x[0], x[1], x[2]
    mx, mn = torch.maximum(r, torch.maximum(g, b)), torch.minimum(r, torch.minimum(g, b))
    d = mx - mn
    safe = torch.where(d > 0, d, torch.ones_like(d))
    hr = (g - b) / safe + torch.where(g < b, 6.0, 0.0)
    hg = (b - r) / safe + 2.0
    hb = (r - g) / safe + 4.0
    hh = torch.where(mx == r, hr, torch.where(mx == g, hg, hb)) * (1.0 / 6.0)
    hh = torch.where(d > 0, hh, torch.zeros_like(hh))
    s = torch.where(mx > 0, d / torch.where(mx > 0, mx, torch.ones_like(mx)), torch.zeros_like(mx))
    v = mx
    hh = hh + np.float32(shift)
    hh = hh - torch.floor(hh)
    h6 = hh * 6.0
    i = torch.floor(h6).long() % 6
    f = h6 - torch.floor(h6)
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    sel = lambda opts: sum(torch.where(i == k, o, torch.zeros_like(o)) for k, o in enumerate(opts))
    return torch.stack([sel((v, q, p, p, t, v)), sel((t, v, v, q, p, p)), sel((p, p, t, v, v, q))])


def _jitter(x, color, mean_grey):
    order, b, c, s, hue = color
    bright_done = False
    for op in order:
        if op == 0:
            x = (x * np.float32(b)).clamp(0, 1)
            bright_done = True
        elif op == 1:
            m = mean_grey * (np.float32(b) if bright_done else np.float32(1.0))
            x = ((x - m) * np.float32(c) + m).clamp(0, 1)
        elif op == 2:
            gy = _grey(x)[None]
            x = ((x - gy) * np.float32(s) + gy).clamp(0, 1)
        else:
            x = _hue_shift(x, hue)
    return x


def apply_plan_host(img_hwc_u8, plan):
    """numpy (h, w, 3) uint8 -> torch (3, S, S) float32 in [0, 1] on the host (the arithmetic of cape_augment_batch)."""
    src = torch.from_numpy(np.ascontiguousarray(img_hwc_u8)).permute(2, 0, 1).float()
    _, h, w = src.shape
    assert (h, w) == (plan.h, plan.w)
    S = plan.out_size
    # 1. warp at crop resolution: aug pixel index -> source pixel index, bilinear, zero padding
    M = torch.tensor(plan.inv[:2], dtype=torch.float32)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    sx = M[0, 0] * xs + M[0, 1] * ys + M[0, 2]
    sy = M[1, 0] * xs + M[1, 1] * ys + M[1, 2]
    grid = torch.stack([(sx + 0.5) / w * 2 - 1, (sy + 0.5) / h * 2 - 1], -1)[None]
    aug = F.grid_sample(src[None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0] * (1.0 / 255.0)
    mean_grey = _grey(aug).mean()
    # 2. colour jitter, 3. blur / noise, at crop resolution
    x = _jitter(aug, plan.color, mean_grey) if plan.color is not None else aug
    if plan.mode == 2:
        k = plan.blur_kernel.shape[0]
        pad = k // 2
        xp = x
        if pad:
            # reflect-101; crops narrower than the kernel radius fall back to the index form the kernel uses
            if min(h, w) > pad:
                xp = F.pad(x[None], (pad, pad, pad, pad), mode="reflect")[0]
            else:
                def refl(i, n):
                    i = np.asarray(i)
                    if n == 1:
                        return np.zeros_like(i)
                    while ((i < 0) | (i >= n)).any():
                        i = np.where(i < 0, -i, i); i = np.where(i >= n, 2 * (n - 1) - i, i)
                    return i
                yi, xi = refl(np.arange(-pad, h + pad), h), refl(np.arange(-pad, w + pad), w)
                xp = x[:, torch.from_numpy(yi)][:, :, torch.from_numpy(xi)]
        ker = torch.from_numpy(plan.blur_kernel)[None, None].expand(3, 1, k, k).contiguous()
        x = F.conv2d(xp[None], ker, groups=3)[0]
    elif plan.mode == 1:
        x = (x + np.float32(plan.noise_std) * _gauss_noise(plan.noise_seed, h, w)).clamp(0, 1)
    # 4. resize: cv2 INTER_LINEAR (half-pixel centres, edge clamp, no antialiasing)
    return F.interpolate(x[None], size=(S, S), mode="bilinear", align_corners=False, antialias=False)[0]


def images_from_raw_host(crops_u8, plans, mean=None, std=None):
    """Host counterpart of DeviceImagePipeline.__call__ (CPU tests; CAPE_HOST_AUGMENT=1): (N, 3, S, S) fp32."""
    out = torch.stack([apply_plan_host(c.numpy() if isinstance(c, torch.Tensor) else c, p) for c, p in zip(crops_u8, plans)])
    if mean is not None:
        out = (out - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    return out


# ------------------------------------------------------------------------------------------------
# GPU implementation
# ------------------------------------------------------------------------------------------------
class DeviceImagePipeline:
    """GPU side of the loader: raw uint8 crops + plans -> a normalised (N, 3, S, S) fp32 batch on `device`, made by
    `cape_augment_batch` on the pipeline's own stream so that the copy / warp of batch i + 1 overlaps the training step of
    batch i.  One pinned staging buffer carries all crops of a batch (one host-to-device copy), a second one the item table."""

    def __init__(self, device, out_size=512, mean=None, std=None):
        self.device, self.S = torch.device(device), out_size
        if self.device.type != "cuda":
            raise RuntimeError("DeviceImagePipeline runs on the GPU (use apply_plan_host on the host)")
        self.stream = torch.cuda.Stream(device=self.device)
        self.mean = None if mean is None else torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std = None if std is None else torch.tensor(std, dtype=torch.float32, device=self.device)

    def __call__(self, crops_u8, plans, wait=True):
        from ..hip import lib, ops
        n = len(crops_u8)
        out = torch.empty(n, 3, self.S, self.S, dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        sizes = [int(c.shape[0]) * int(c.shape[1]) for c in crops_u8]
        off = np.concatenate([[0], np.cumsum([(s * 3 + 15) // 16 * 16 for s in sizes])]).astype(np.int64)
        stage = torch.empty(int(off[-1]), dtype=torch.uint8).pin_memory()
        for c, o, s in zip(crops_u8, off[:-1], sizes):
            t = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c))
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
            stage[o:o + 3 * s].copy_(t.reshape(-1))
        foff = np.concatenate([[0], np.cumsum([s * 3 for s in sizes])]).astype(np.int64)
        with torch.cuda.stream(self.stream):
            src = stage.to(self.device, non_blocking=True)
            work = torch.empty(int(foff[-1]), dtype=torch.float32, device=self.device)
            stat = torch.zeros(n, dtype=torch.float32, device=self.device)
            items = (lib.AugItem * n)()
            for i, (c, p) in enumerate(zip(crops_u8, plans)):
                it = items[i]
                h, w = int(c.shape[0]), int(c.shape[1])
                assert (h, w) == (p.h, p.w) and p.out_size == self.S
                it.src, it.aug = src.data_ptr() + int(off[i]), work.data_ptr() + 4 * int(foff[i])
                it.out, it.stat = out.data_ptr() + 4 * i * 3 * self.S * self.S, stat.data_ptr() + 4 * i
                it.h, it.w = h, w
                for k, v in enumerate(p.inv[:2].reshape(-1)):
                    it.M[k] = float(v)
                it.color_on = int(p.color is not None)
                if p.color is not None:
                    order, it.bright, it.contrast, it.sat, it.hue = p.color
                    for k in range(4):
                        it.order[k] = int(order[k])
                it.mode, it.noise_std, it.seed = p.mode, p.noise_std, p.noise_seed
                if p.mode == 2:
                    kk = int(p.blur_kernel.shape[0])
                    assert kk * kk <= 49
                    it.blur_k = kk
                    for k, v in enumerate(p.blur_kernel.reshape(-1)):
                        it.blur_w[k] = float(v)
            tab_host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).pin_memory()
            tab = tab_host.to(self.device, non_blocking=True)
            ops.augment_batch(tab, n, max(sizes), self.S, self.mean, self.std)
            self._keep = (stage, tab_host, src, work, stat, tab)       # alive until the next batch is staged (stream order)
            for t in (src, work, stat, tab, out):
                t.record_stream(torch.cuda.current_stream(self.device))
        if wait:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out


class HostTransform:
    """Callable with the albumentations contract the dataset uses: `t(image=hwc_uint8, keypoints=[(x, y)...])` ->
    {'image': (S, S, 3) uint8-range float array, 'keypoints': [...]} (`mp100_cape.py:566-577`).

    The plan generator is seeded lazily, in the process that draws from it: (seed, rank, DataLoader worker seed).  torch reseeds
    every worker per epoch (base_seed + worker_id), so two workers, two ranks and two epochs draw different augmentation streams
    -- one generator created in the parent and inherited by every forked worker would repeat the same plans everywhere."""

    def __init__(self, train=False, size=512, seed=None, rank=0):
        self.train, self.size, self.seed, self.rank = train, size, seed, rank
        self._rng, self._rng_key = None, None

    @property
    def rng(self):
        info = torch.utils.data.get_worker_info()
        key = (os.getpid(), None if info is None else info.seed)
        if self._rng is None or key != self._rng_key:
            ent = [0 if self.seed is None else int(self.seed) & 0xFFFFFFFF, int(self.rank)]
            if info is not None:
                ent += [int(info.seed) & 0xFFFFFFFF, int(info.seed) >> 32]
            self._rng = np.random.default_rng(np.random.SeedSequence(ent) if (self.seed is not None or info is not None) else None)
            self._rng_key = key
        return self._rng

    def plan(self, h, w):
        return train_plan(h, w, self.rng, self.size) if self.train else resize_plan(h, w, self.size)

    def __call__(self, image, keypoints):
        h, w = image.shape[:2]
        plan = self.plan(h, w)
        out = apply_plan_host(image, plan)
        return {"image": (out.permute(1, 2, 0).numpy() * 255.0), "keypoints": plan.map_keypoints(keypoints), "plan": plan}
