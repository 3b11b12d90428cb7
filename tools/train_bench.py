import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import invr
from invr import scene, _abi
from invr.config import make_cfg
from invr.network import Network
from invr.trainer import NetworkWrapper
DEV='cuda:0'
cfg = make_cfg(N_samples=64)
with torch.device(DEV): net = Network(cfg=cfg)
net = net.to(DEV).train()
g = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for name, p in net.named_parameters():
        if name.endswith('embedder.dense') or name.endswith('embedder.hash'): p.normal_(0.0, 0.01, generator=g)
bnp,_ = scene.make_scene(256,256,seed=0,cam_dist=1.8, crop=(96,96,64,64))
gb = {k:v.to(DEV) for k,v in scene.to_torch(bnp).items()}
print('patch rays', gb['ray_o'].shape[1])
wrap = NetworkWrapper(net)
groups = [{'params':[p],'lr':5e-4} for p in net.parameters() if p.requires_grad]
from invr import driver
opt = driver.make_optimizer(net, fused=not os.environ.get('TORCH_ADAM'))
def step(i, timing=None):
    gb['iter_step'] = i+2
    t0=time.perf_counter(); ret, loss, stats, _ = wrap(gb, split='train'); loss = loss.mean(); torch.cuda.synchronize(); t1=time.perf_counter()
    opt.zero_grad(set_to_none=True); loss.backward(); torch.cuda.synchronize(); t2=time.perf_counter()
    opt.step(); torch.cuda.synchronize(); t3=time.perf_counter()
    if timing is not None: timing.append((t1-t0,t2-t1,t3-t2))
    return float(loss)
for i in range(3): l=step(i)
T=[]
for i in range(10): l=step(i+3,T)
T=np.array(T)*1e3
print('loss',l,'fwd/bwd/opt ms', T.mean(0), 'total', T.sum(1).mean())
print('stats', wrap.renderer.last_stats.cpu().numpy()[:8])
if os.environ.get('PROF'):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(3): step(20 + i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by=os.environ.get('PROF_SORT', 'cuda_time_total'), row_limit=28, max_name_column_width=60))
