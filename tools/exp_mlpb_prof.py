"""Phase timers of k_part_mlp_bwd (variants/libinvr_mlpbprof.so, built with -DMLPB_PROF): cycles per wave in staging / input loads /
forward recompute / backward head .. rgb2^T / rgb1^T .. occ1^T + stores / tail, over a few training iterations."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['INVR_LIB_PATH'] = os.path.join(R, 'variants', 'libinvr_mlpbprof.so')
sys.path.insert(0, R)
import torch
import invr  # noqa
from invr import _abi, scene, driver
from invr.config import make_cfg
from invr.network import Network
from invr.trainer import NetworkWrapper
DEV = 'cuda:0'
cfg = make_cfg(N_samples=128)
with torch.device(DEV):
    net = Network(cfg=cfg)
net = net.to(DEV).train()
bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8, crop=(240, 240, 32, 32))
gb = {k: v.to(DEV) for k, v in scene.to_torch(bnp).items()}
wrap = NetworkWrapper(net)
opt = driver.make_optimizer(net)
fn = _abi.lib().invr_debug_mlpb_prof
fn.argtypes = [C.c_void_p, C.c_int]
for i in range(4):
    driver.train_step(wrap, opt, gb, i + 2)
torch.cuda.synchronize()
fn(None, 1)
N = 5
for i in range(N):
    driver.train_step(wrap, opt, gb, i + 6)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
assert fn(buf, 0) == 0
v = [int(x) for x in buf]
waves = max(v[8], 1)
names = ['0 weight staging + barrier', '1 input loads (waited)', '2 forward recompute', '3 backward: head, rgb2^T, stores', '4 rgb1^T, occ2^T, occ1^T, stores', '5 tail (latent reduction)', '6 -', '7 loop head']
tot = sum(v[:8])
print('waves with work (all parts, %d iterations): %d' % (N, waves))
for i, nme in enumerate(names):
    print('%-40s %10.0f cycles per wave  %5.1f %%' % (nme, v[i] / waves, 100.0 * v[i] / max(tot, 1)))
print('total %.0f cycles per wave' % (tot / waves))
