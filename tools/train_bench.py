"""Training-iteration timing on one MI355X (BASELINE configs[4] shape by default: 1024 rays x 128 samples, full-size model).
  python tools/train_bench.py [--rays-side 32] [--samples 128] [--iters 50] [--graph] [--prof]
Reports the un-synchronised iteration time (the loop only syncs at the end) and a synchronised fwd / bwd / opt split."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import invr  # noqa: E402,F401
from invr import scene, driver  # noqa: E402
from invr.config import make_cfg  # noqa: E402
from invr.network import Network  # noqa: E402
from invr.trainer import NetworkWrapper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rays-side', type=int, default=32)
ap.add_argument('--samples', type=int, default=128)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--graph', action='store_true', help='op-by-op autograd graph (cfg.train_fused False)')
ap.add_argument('--torch-adam', action='store_true')
ap.add_argument('--prof', action='store_true')
ap.add_argument('--thresh', type=float, default=0.05)
args = ap.parse_args()
DEV = 'cuda:0'
cfg = make_cfg(N_samples=args.samples, smpl_thresh=args.thresh)
cfg.train_fused = not args.graph
with torch.device(DEV):
    net = Network(cfg=cfg)
net = net.to(DEV).train()
g = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for name, p in net.named_parameters():
        if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
            p.normal_(0.0, 0.1, generator=g)
s = args.rays_side
bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8, crop=(256 - s // 2, 256 - s // 2, s, s))
gb = {k: v.to(DEV) for k, v in scene.to_torch(bnp).items()}
print('patch rays', gb['ray_o'].shape[1], 'samples', args.samples)
wrap = NetworkWrapper(net)
opt = driver.make_optimizer(net, fused=not args.torch_adam)


def step(i, timing=None):
    gb['iter_step'] = i + 2
    if timing is None:
        return driver.train_step(wrap, opt, gb, i + 2)[0]
    t0 = time.perf_counter()
    ret, loss, stats, _ = wrap(gb, split='train'); loss = loss.mean(); torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True); loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
    timing.append((t1 - t0, t2 - t1, t3 - t2))
    return loss.detach()


for i in range(5):
    l = step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.iters):
    l = step(i + 5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
print('iteration %.3f ms (un-synchronised loop), %.3e ray-samples/s, loss %.5f' % (dt * 1e3, gb['ray_o'].shape[1] * args.samples / dt, float(l)))
T = []
for i in range(10):
    step(100 + i, T)
T = np.array(T) * 1e3
print('synchronised fwd / bwd / opt ms', T.mean(0), 'sum', T.sum(1).mean())
print('stats', wrap.renderer.last_stats.cpu().numpy()[:15])
if args.prof:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(5):
            step(200 + i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=70))
if os.environ.get('HOSTPROF'):
    # host side of the loop: time to ISSUE n iterations (no synchronisation inside) vs time until the GPU has finished them, and a cProfile
    # of the issuing thread
    import cProfile, pstats, io
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        step(300 + i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_done = time.perf_counter() - t0
    print('40 iterations: issued in %.3f ms each, finished in %.3f ms each' % (t_issue / 40 * 1e3, t_done / 40 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(40):
        step(400 + i)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
    print(s.getvalue()[:9000])
