"""The bench's api_frame probe in its own order (to_cpu False first, after empty_cache): per-sweep and slowest-call times."""
import os, sys, time
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
torch.cuda.synchronize(); torch.cuda.empty_cache()
r = Renderer(net); r.in_flight = 8
for to_cpu in (False, True, False):
    r.eval_to_cpu = to_cpu
    q = deque()
    def sweep(n):
        slow = []
        for i in range(n):
            t = time.perf_counter()
            q.append(r.render(dict(batches[i % 10])))
            while len(q) >= r.in_flight:
                o = q.popleft(); _ = o['rgb_map'], o['acc_map']
            dt = (time.perf_counter() - t) * 1e3
            if dt > 6: slow.append((i, round(dt, 1)))
        while q:
            o = q.popleft(); _ = o['rgb_map'], o['acc_map']
        torch.cuda.synchronize()
        return slow
    for rep in range(4):
        t0 = time.perf_counter(); slow = sweep(40); dt = (time.perf_counter() - t0) / 40 * 1e3
        print('to_cpu', to_cpu, 'sweep', rep, 'ms per frame %.3f' % dt, 'reserved GB %.1f' % (torch.cuda.memory_reserved() / 1e9), 'cap_hint', r._cap_hint, 'slow calls', slow[:12], flush=True)
