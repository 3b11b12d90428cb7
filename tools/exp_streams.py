"""Experiment: K frames in flight as K branches of one hipGraph replay against K eager launch chains on K streams (no graph:
the shape Renderer.render needs when every batch is a new dict with its own volume dimensions).
python tools/exp_streams.py [--shard-of W] [K ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from invr.config import make_cfg

args = sys.argv[1:]
W = 0
if args and args[0] == '--shard-of':
    W = int(args[1]); args = args[2:]
dev = torch.device('cuda', 0)
S = 128
cfg = make_cfg(N_samples=S)
net = bench.build_model(cfg, dev)
Ks = [int(a) for a in args] or [1, 2, 4, 8]
_, batches = bench.frame_batches(512, 1.8, max(Ks), dev)
for K in Ks:
    for mode in ('graph', 'streams'):
        fs = bench.frame_set(net, batches[:K], S, 0, 1, shard_of=W, capture=mode == 'graph', streams=mode == 'streams')
        ms = bench.time_frames(fs.replay, max(1, 40 // K), 0.5) / K
        print('shard 1/%d, K = %2d, %-7s: %.3f ms per frame' % (max(W, 1), K, mode, ms), flush=True)
        del fs
        torch.cuda.empty_cache()
