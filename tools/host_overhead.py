import sys, time, torch
sys.path.insert(0,'/root/repo')
import cape_amd
from cape_amd.hip import ops, functional as HF
x=torch.randn(64,256,device='cuda'); w=torch.randn(256,256,device='cuda'); b=torch.randn(256,device='cuda'); y=torch.empty(64,256,device='cuda')
def t(fn,n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    dt=time.perf_counter()-t0; torch.cuda.synchronize(); return dt/n*1e6
print("ops.gemm            %.1f us"%t(lambda: ops.gemm(x,w,y,64,256,256,bias=b)))
print("ops.add             %.1f us"%t(lambda: ops.add(x,x)))
print("torch.add           %.1f us"%t(lambda: torch.add(x,x)))
print("torch.empty         %.1f us"%t(lambda: torch.empty(64,256,device='cuda')))
xr=x.clone().requires_grad_(True); wr=w.clone().requires_grad_(True); br=b.clone().requires_grad_(True)
print("HF.linear fwd       %.1f us"%t(lambda: HF.linear(xr,wr,br)))
def fb():
    o=HF.linear(xr,wr,br); o.backward(x)
print("HF.linear fwd+bwd   %.1f us"%t(fb,500))
print("add_layernorm fwd   %.1f us"%t(lambda: ops.add_layernorm_fwd(x,x,b,b)))
import cProfile,pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): ops.gemm(x,w,y,64,256,256,bias=b)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(12)
from cape_amd.runtime.arena import ParamGroupArena  # noqa
lin = torch.nn.Linear(256, 256).cuda()
from cape_amd.runtime.optimizer import ArenaAdamW
class M(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.l = lin
opt = ArenaAdamW(M().cuda(), lr=1e-4, lr_backbone=1e-5)
def fb2():
    o = HF.linear(xr, lin.weight, lin.bias); o.backward(x)
print("HF.linear fwd+bwd (arena, side stream) %.1f us" % t(fb2, 500))
pr=cProfile.Profile(); pr.enable()
for _ in range(500): fb2()
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(22)
