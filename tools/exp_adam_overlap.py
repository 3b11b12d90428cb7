"""Experiment: the optimizer step of the part TABLES (6.5 of the 6.9 GB k_adam moves) on a side stream beside the deformer stage of the
backward, the rest at optimizer.step() — timing only.  python tools/exp_adam_overlap.py [--iters 40] [--streams N: try N side streams]"""
import argparse, os, sys, time
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import invr  # noqa
from invr import scene, driver, _abi
from invr.config import make_cfg
from invr.network import Network
from invr.trainer import NetworkWrapper

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=40)
ap.add_argument('--streams', type=int, default=4)
args = ap.parse_args()
DEV = 'cuda:0'
cfg = make_cfg(N_samples=128)
with torch.device(DEV):
    net = Network(cfg=cfg)
net = net.to(DEV).train()
g = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for name, p in net.named_parameters():
        if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
            p.normal_(0.0, 0.1, generator=g)
bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8, crop=(240, 240, 32, 32))
gb = {k: v.to(DEV) for k, v in scene.to_torch(bnp).items()}
wrap = NetworkWrapper(net)
opt = driver.make_optimizer(net)
L = _abi.lib()


def loop(n, tag):
    for i in range(5):
        driver.train_step(wrap, opt, gb, i + 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        driver.train_step(wrap, opt, gb, i + 7)
    torch.cuda.synchronize()
    print('%-46s %.3f ms per iteration' % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)


loop(args.iters, 'baseline (k_adam behind the backward)')
loop(args.iters, 'baseline again')

# split the optimizer's chunk list: part tables / the rest
ct = opt._chunk_tensor.cpu().numpy()
ci = opt._chunk_index.cpu().numpy()
is_tab = np.array([id(p) in opt._row_grad_of for p in opt._plan_params])
m = is_tab[ct]
dev = opt._chunk_tensor.device
tab_ct, tab_ci = torch.from_numpy(ct[m]).to(dev), torch.from_numpy(ci[m]).to(dev)
rest_ct, rest_ci = torch.from_numpy(ct[~m]).to(dev), torch.from_numpy(ci[~m]).to(dev)
print('chunks: tables %d, rest %d' % (tab_ct.numel(), rest_ct.numel()))
betas, eps = opt.param_groups[0]['betas'], opt.param_groups[0]['eps']
n_ent = len(opt._plan_params)


class Beside:
    part_order = [0]
    def __init__(self, side):
        self.side = side
    def reduce_part(self, p):
        cur = torch.cuda.current_stream()
        _abi.check(L.invr_adam_advance(C.c_void_p(opt._table.data_ptr()), n_ent, betas[0], betas[1], _abi.stream_ptr()))
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            _abi.check(L.invr_adam_step(C.c_void_p(opt._table.data_ptr()), _abi.ptr(tab_ct, torch.int32), _abi.ptr(tab_ci, torch.int32), tab_ct.numel(),
                                        betas[0], betas[1], eps, _abi.stream_ptr()))
    def reduce_small(self):
        pass
    def wait(self):
        pass


def step_rest(closure=None):
    _abi.check(L.invr_adam_step(C.c_void_p(opt._table.data_ptr()), _abi.ptr(rest_ct, torch.int32), _abi.ptr(rest_ci, torch.int32), rest_ct.numel(),
                                betas[0], betas[1], eps, _abi.stream_ptr()))
    torch.cuda.current_stream().wait_stream(opt.arena.reducer.side)
    torch.autograd.graph.increment_version(opt._plan_params)


orig_step = opt.step
opt.step = step_rest
opt._replicas_checked = True
streams = [torch.cuda.Stream(DEV) for _ in range(args.streams)]
for k, s in enumerate(streams):
    opt.arena.reducer = Beside(s)
    loop(args.iters, 'tables beside the deformer stage, side stream %d' % k)
