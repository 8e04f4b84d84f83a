"""Image / keypoint transforms of the MP-100 loader, MI355X-first.

The reference pipes every crop through albumentations on the host (`datasets/mp100_cape.py:888-950`: train = Affine
(translate +-10 %, scale 0.85-1.15, rotate +-30 deg, p 0.7) + HorizontalFlip(0.5) + ColorJitter(0.3, 0.3, 0.3, 0.1, p 0.6) +
OneOf(noise / blur, p 0.3) + Resize(512); val / test = Resize(512)).  Here a transform is a small *plan* drawn on the host
(one 2x3 affine map output pixel -> source pixel that already contains resize, rotation, scale, shift and flip, plus
brightness / contrast / saturation factors and a noise seed); keypoints follow the inverse map on the host (a few numbers),
pixels are produced either
  * on the host (`apply_plan_host`, torch bilinear sampling: DataLoader workers, the CPU tests), or
  * on the GPU (`DeviceImagePipeline`): workers ship the raw uint8 crop (variable size, pinned), one HIP-side pass per image
    (`torch.nn.functional.grid_sample` on the copy stream today) warps, resizes, jitters and normalises into the (3, S, S)
    fp32 batch slot, so the host cores only decode and crop and do not cap episodes/s at 8 GPUs.
Both produce the same pixels for the same plan (bilinear, zero padding, half-pixel centres: albumentations' cv2
INTER_LINEAR / BORDER_CONSTANT convention; cv2's fixed-point uint8 rounding is not reproduced -- parity of augmented pixels
is unpinned, albumentations is not installed; the geometry of keypoints is exact)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


class TransformPlan:
    """out pixel (u, v) samples source (x, y) = M @ (u + 0.5, v + 0.5, 1) - 0.5 ; keypoints move by the inverse."""

    def __init__(self, M, out_size, brightness=1.0, contrast=1.0, saturation=1.0, noise_std=0.0, noise_seed=0, flipped=False):
        self.M = np.asarray(M, dtype=np.float64).reshape(2, 3)
        self.out_size = int(out_size)
        self.brightness, self.contrast, self.saturation = float(brightness), float(contrast), float(saturation)
        self.noise_std, self.noise_seed, self.flipped = float(noise_std), int(noise_seed), bool(flipped)

    def map_keypoints(self, kpts):
        A = np.vstack([self.M, [0.0, 0.0, 1.0]])
        Ainv = np.linalg.inv(A)
        k = np.asarray(kpts, dtype=np.float64).reshape(-1, 2)
        out = (np.c_[k, np.ones(len(k))] @ Ainv.T)[:, :2]
        return [(float(x), float(y)) for x, y in out]


def resize_plan(h, w, size=512):
    """albumentations.Resize(size, size): keypoints scale by size/w, size/h (`mp100_cape.py:942-944`)."""
    return TransformPlan([[w / size, 0.0, 0.0], [0.0, h / size, 0.0]], size)


def train_plan(h, w, rng, size=512):
    """Random plan with the reference's training distribution (`mp100_cape.py:896-931`), drawn from `rng` (numpy Generator)."""
    sx, sy = w / size, h / size
    # resize then (in 512-space) affine about the image centre, then optional horizontal flip
    A = np.eye(3)
    if rng.random() < 0.7:
        ang = math.radians(rng.uniform(-30, 30))
        sc = rng.uniform(0.85, 1.15)
        tx, ty = rng.uniform(-0.1, 0.1) * size, rng.uniform(-0.1, 0.1) * size
        c = size / 2.0
        R = np.array([[sc * math.cos(ang), -sc * math.sin(ang), 0.0], [sc * math.sin(ang), sc * math.cos(ang), 0.0], [0.0, 0.0, 1.0]])
        T0, T1 = np.array([[1, 0, -c], [0, 1, -c], [0, 0, 1.0]]), np.array([[1, 0, c + tx], [0, 1, c + ty], [0, 0, 1.0]])
        A = T1 @ R @ T0                       # forward map in output space: p_out = A p_resized
    flipped = rng.random() < 0.5
    if flipped:
        A = np.array([[-1, 0, size], [0, 1, 0], [0, 0, 1.0]]) @ A
    S = np.array([[sx, 0, 0], [0, sy, 0], [0, 0, 1.0]])          # resized -> source
    M = (S @ np.linalg.inv(A))[:2]
    b = c_ = s_ = 1.0
    if rng.random() < 0.6:
        b, c_, s_ = rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3), rng.uniform(0.7, 1.3)
    noise = rng.uniform(0.01, 0.04) if rng.random() < 0.1 else 0.0
    return TransformPlan(M, size, b, c_, s_, noise, int(rng.integers(1 << 31)), flipped)


def _sample(img_chw_float, plan, device):
    """Bilinear warp of a (3, h, w) float image in [0, 1] to (3, S, S) by the plan (zero padding outside)."""
    _, h, w = img_chw_float.shape
    S = plan.out_size
    u = torch.arange(S, dtype=torch.float32, device=device) + 0.5
    vv, uu = torch.meshgrid(u, u, indexing="ij")
    M = torch.tensor(plan.M, dtype=torch.float32, device=device)
    x = M[0, 0] * uu + M[0, 1] * vv + M[0, 2]                    # source coordinates in pixel-centre convention (+0.5 kept)
    y = M[1, 0] * uu + M[1, 1] * vv + M[1, 2]
    grid = torch.stack([x / w * 2 - 1, y / h * 2 - 1], -1)[None]
    out = F.grid_sample(img_chw_float[None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0]
    if plan.brightness != 1.0 or plan.contrast != 1.0 or plan.saturation != 1.0:
        out = out * plan.brightness
        gray = (0.299 * out[0] + 0.587 * out[1] + 0.114 * out[2])
        out = (out - gray.mean()) * plan.contrast + gray.mean()
        gray = (0.299 * out[0] + 0.587 * out[1] + 0.114 * out[2])[None]
        out = (out - gray) * plan.saturation + gray
    if plan.noise_std > 0:
        g = torch.Generator(device="cpu").manual_seed(plan.noise_seed)
        out = out + torch.randn(out.shape, generator=g).to(device) * plan.noise_std
    return out.clamp_(0.0, 1.0)


def apply_plan_host(img_hwc_u8, plan):
    """numpy (h, w, 3) uint8 -> torch (3, S, S) float32 in [0, 1] on the host."""
    t = torch.from_numpy(np.ascontiguousarray(img_hwc_u8)).permute(2, 0, 1).float() / 255.0
    return _sample(t, plan, "cpu")


class DeviceImagePipeline:
    """GPU side of the loader: raw uint8 crops + plans -> a normalised (N, 3, S, S) fp32 batch on `device`, on its own
    stream so that the copy / warp of batch i+1 overlaps the training step of batch i."""

    def __init__(self, device, out_size=512, mean=None, std=None):
        self.device, self.S = torch.device(device), out_size
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.mean = None if mean is None else torch.tensor(mean, device=self.device).view(3, 1, 1)
        self.std = None if std is None else torch.tensor(std, device=self.device).view(3, 1, 1)

    def __call__(self, crops_u8, plans):
        out = torch.empty(len(crops_u8), 3, self.S, self.S, dtype=torch.float32, device=self.device)
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _Null()
        with ctx:
            for i, (c, p) in enumerate(zip(crops_u8, plans)):
                t = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c))
                if self.device.type == "cuda":
                    t = t.pin_memory().to(self.device, non_blocking=True)
                img = _sample(t.permute(2, 0, 1).float() / 255.0, p, self.device)
                if self.mean is not None:
                    img = (img - self.mean) / self.std
                out[i] = img
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class HostTransform:
    """Callable with the albumentations contract the dataset uses: `t(image=hwc_uint8, keypoints=[(x, y)...])` ->
    {'image': (S, S, 3) uint8-range float array, 'keypoints': [...]} (`mp100_cape.py:566-577`)."""

    def __init__(self, train=False, size=512, seed=None):
        self.train, self.size = train, size
        self.rng = np.random.default_rng(seed)

    def plan(self, h, w):
        return train_plan(h, w, self.rng, self.size) if self.train else resize_plan(h, w, self.size)

    def __call__(self, image, keypoints):
        h, w = image.shape[:2]
        plan = self.plan(h, w)
        out = apply_plan_host(image, plan)
        return {"image": (out.permute(1, 2, 0).numpy() * 255.0), "keypoints": plan.map_keypoints(keypoints), "plan": plan}
