import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import deque
import bench
from invr.config import make_cfg
from invr.renderer import Renderer
dev = torch.device('cuda', 0)
S = 128
cfg = make_cfg(N_samples=S)
net = bench.build_model(cfg, dev)
_, batches = bench.frame_batches(512, 1.8, 10, dev)
for to_cpu, pin in ((True, True), (False, True)):
    r = Renderer(net); r.eval_to_cpu, r.in_flight, r.pin_host = to_cpu, 8, pin
    q = deque()
    def sweep(n):
        for i in range(n):
            q.append(r.render(dict(batches[i % 10])))
            if len(q) >= 8:
                o = q.popleft(); _ = o['rgb_map'], o['acc_map']
        while q:
            o = q.popleft(); _ = o['rgb_map'], o['acc_map']
        torch.cuda.synchronize()
    sweep(30)
    t0 = time.perf_counter(); sweep(40); dt = (time.perf_counter() - t0) / 40 * 1e3
    print('to_cpu', to_cpu, 'ms per frame', dt, 'allocated GB', torch.cuda.memory_allocated() / 1e9, 'reserved GB', torch.cuda.memory_reserved() / 1e9, flush=True)
    pr = cProfile.Profile(); pr.enable(); sweep(40); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(4)
    r.flush(release=True); del r; torch.cuda.empty_cache()
