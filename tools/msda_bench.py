"""Isolated timing of the MSDA backward forms on the encoder / decoder shapes of the CAPE training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402

SHAPES = [(32, 32), (16, 16), (8, 8), (4, 4)]


def main():
    R = int(os.environ.get("IMAGE", "256"))
    shapes = [(R // 8, R // 8), (R // 16, R // 16), (R // 32, R // 32), (R // 64, R // 64)]
    geo = ops.LevelGeometry(shapes)
    for N, Lq in ((32, geo.S), (32, 40), (8, geo.S)):
        g = torch.Generator(device="cuda").manual_seed(0)
        value = torch.randn(N, geo.S, 256, device="cuda", generator=g)
        off = torch.randn(N, Lq, 256, device="cuda", generator=g) * float(os.environ.get("OFF_SCALE", "1.5"))
        offw = torch.cat([off, torch.randn(N, Lq, 128, device="cuda", generator=g)], -1).contiguous()
        ref = torch.rand(N, Lq, 4, 2, device="cuda", generator=g)
        go = torch.randn(N, Lq, 256, device="cuda", generator=g)
        for _ in range(3):
            ops.msda_fwd(value, offw, ref, geo, N, Lq)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.msda_fwd(value, offw, ref, geo, N, Lq)
        e1.record()
        torch.cuda.synchronize()
        print(f"N={N} Lq={Lq} forward: {e0.elapsed_time(e1) / 20 * 1e3:9.1f} us", flush=True)
        for form in ("atomic", "f64", "fx"):
            for need_ref in (False, True):
                for _ in range(3):
                    ops.msda_bwd(go, value, offw, ref, geo, N, Lq, need_ref_grad=need_ref, form=form)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                it = 10
                e0.record()
                for _ in range(it):
                    ops.msda_bwd(go, value, offw, ref, geo, N, Lq, need_ref_grad=need_ref, form=form)
                e1.record()
                torch.cuda.synchronize()
                print(f"N={N} Lq={Lq} form={form} d_ref={need_ref}: {e0.elapsed_time(e1) / it * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
