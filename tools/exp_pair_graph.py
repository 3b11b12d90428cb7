"""Experiment: K frames of a 1/W shard as parallel branches of ONE captured hipGraph (ROCm runs branches of a graph side by side,
separate graph launches on separate streams did not overlap: gpurun_out/r4b).  usage: python tools/exp_pair_graph.py W K"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from invr import scene as scene_mod, dist as idist
from invr.config import make_cfg
W, K = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
cfg = make_cfg(N_samples=128)
net = bench.build_model(cfg, dev)
bnp, _ = scene_mod.make_scene(512, 512, seed=0, cam_dist=1.8)
batch = {k: v.to(dev) for k, v in scene_mod.to_torch(bnp).items()}
n_rays = batch['ray_o'].shape[1]
idx = idist.tile_indices(n_rays, 0, W, device=dev)
a = tuple(batch[k][0][idx].contiguous() for k in ('ray_o', 'ray_d', 'near', 'far'))
ctx = net.prepare(batch)
wss = []
def render(k):
    net._ws = wss[k] if k < len(wss) else None
    out = net.render_rays(ctx, a[0], a[1], a[2], a[3], 128, want_raw=True)
    if k >= len(wss): wss.append(net._ws)
    return out
for k in range(K): render(k)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
streams = [torch.cuda.Stream() for _ in range(K)]
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    cur = torch.cuda.current_stream()
    outs = []
    for k in range(K):
        if k == 0:
            outs.append(render(0))
        else:
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                outs.append(render(k))
    for k in range(1, K): cur.wait_stream(streams[k])
ms = bench.time_frames(g.replay, 20, 0.5)
print('W=%d: %d frames per graph replay: %.4f ms per replay = %.4f ms per frame' % (W, K, ms, ms / K))
for k in range(1, K): assert torch.equal(outs[0]['rgb_map'], outs[k]['rgb_map'])
