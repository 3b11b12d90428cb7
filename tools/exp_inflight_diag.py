"""Renderer.render with frames in flight across calls: ms per frame, allocator footprint, and where the host time of a render() call goes
(INVR_DIAG_PHASES=1: per-phase wall-clock inside _render_in_flight, monkey-patched)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import deque, defaultdict
import bench
from invr.config import make_cfg
from invr import renderer as R
from invr.renderer import Renderer
dev = torch.device('cuda', 0)
S = 128
cfg = make_cfg(N_samples=S)
net = bench.build_model(cfg, dev)
_, batches = bench.frame_batches(512, 1.8, 10, dev)
ph = defaultdict(list)
if os.environ.get('INVR_DIAG_PHASES'):
    def timed(obj, name, key):
        f = getattr(obj, name)
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); ph[key].append((time.perf_counter() - t) * 1e3); return r
        setattr(obj, name, g)
    timed(R._PendingFrame, 'result', 'join'); timed(R._PendingFrame, '_run', 'run')
    timed(net, 'prepare', 'prepare'); timed(net, 'render_rays', 'render_rays')
for rep in range(2):
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
    for k in ph: ph[k].clear()
    t0 = time.perf_counter(); sweep(40); dt = (time.perf_counter() - t0) / 40 * 1e3
    print('to_cpu', to_cpu, 'ms per frame %.3f' % dt, 'allocated GB %.1f' % (torch.cuda.memory_allocated() / 1e9), 'reserved GB %.1f' % (torch.cuda.memory_reserved() / 1e9),
          {k: '%.2f / max %.1f' % (sum(v) / max(len(v), 1), max(v) if v else 0) for k, v in ph.items()}, flush=True)
    r.flush(); del r; torch.cuda.synchronize()
