"""Experiment: ONE frame of a 1/W shard rendered as K ray sub-shards on K parallel branches of one captured hipGraph.
usage: python tools/exp_split_graph.py W K [all_side]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from invr import scene as scene_mod, dist as idist
from invr.config import make_cfg
W, K = int(sys.argv[1]), int(sys.argv[2])
all_side = len(sys.argv) > 3
dev = torch.device('cuda', 0)
cfg = make_cfg(N_samples=128)
net = bench.build_model(cfg, dev)
bnp, _ = scene_mod.make_scene(512, 512, seed=0, cam_dist=1.8)
batch = {k: v.to(dev) for k, v in scene_mod.to_torch(bnp).items()}
n_rays = batch['ray_o'].shape[1]
idx = idist.tile_indices(n_rays, 0, W, device=dev)
subs = [idx[idist.tile_indices(idx.numel(), k, K, device=dev)] for k in range(K)]
args = [tuple(batch[k][0][s].contiguous() for k in ('ray_o', 'ray_d', 'near', 'far')) for s in subs]
ctx = net.prepare(batch)
wss = []
def render(k):
    net._ws = wss[k] if k < len(wss) else None
    a = args[k]
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
        if k == 0 and not all_side:
            outs.append(render(0))
        else:
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                outs.append(render(k))
    for k in range(K):
        if k or all_side: cur.wait_stream(streams[k])
ms = bench.time_frames(g.replay, 20, 0.5)
print('W=%d: one frame as %d sub-shard branches%s: %.4f ms per frame' % (W, K, ' (all on side streams)' if all_side else '', ms))
