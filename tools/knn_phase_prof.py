"""Where does a wave of k_knn_pairs spend its time?  Runs bench-shaped frames through the -DKNN_PROF build of the library
(tools/knn_phase_prof.sh) and prints the s_memtime cycles per phase, summed over waves, per wave-tile (64 points)."""
import argparse
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['INVR_LIB_PATH'] = os.path.join(R, 'tools', '_prof', 'libinvr_knnprof.so')
sys.path.insert(0, R)
import torch  # noqa: E402
import invr  # noqa: E402,F401
from invr import _abi, scene  # noqa: E402
from invr.config import make_cfg  # noqa: E402
from invr.network import Network  # noqa: E402
from invr.renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shard-of', type=int, default=1)
ap.add_argument('--frames', type=int, default=5)
args = ap.parse_args()
DEV = 'cuda:0'
cfg = make_cfg(N_samples=128)
with torch.device(DEV):
    net = Network(cfg=cfg)
net = net.to(DEV).eval()
bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8)
gb = {k: v.to(DEV) for k, v in scene.to_torch(bnp).items()}
if args.shard_of > 1:
    from invr import dist as idist
    idx = idist.tile_indices(gb['ray_o'].shape[1], 0, args.shard_of, 512).to(DEV)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        gb[k] = gb[k][:, idx]
rend = Renderer(net)
L = _abi.lib()
fn = L.invr_debug_knn_prof
fn.argtypes = [C.c_void_p, C.c_int]
with torch.no_grad():
    for _ in range(2):
        rend.render(gb)
    torch.cuda.synchronize()
    fn(None, 1)
    for _ in range(args.frames):
        rend.render(gb)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
assert fn(buf, 0) == 0
v = [int(x) for x in buf]
names = ['0 ticket + barrier', '1 sample point, lattice cell', '2 part classification', '3 sweep', '4 top-4 finish + weights', '5 list append', '6 LDS staging (once per workgroup)']
tiles = max(v[8], 1)
print('wave-tiles per frame %.0f; part scans / wave-tile %.2f; clusters visited / scan %.2f; sub-clusters scanned / scan %.2f'
      % (v[8] / args.frames, v[9] / tiles, v[10] / max(v[9], 1), v[11] / max(v[9], 1)))
tot = sum(v[:6])
for i, n in enumerate(names):
    print('%-38s %10.0f cycles / wave-tile  %5.1f %%' % (n, v[i] / tiles, 100.0 * v[i] / max(tot + v[6], 1)))
print('total per wave-tile %.0f cycles (s_memtime ticks: 100 MHz constant clock on gfx9 -> x10 ns)' % (tot / tiles))

n = max(v[16 + 8], 1)
print('k_part_prepare (workgroup 0 = part 0), cycles per launch: AABB %.0f  keys %.0f  sort %.0f  vertex write %.0f  cluster records %.0f'
      % tuple(v[16 + i] / n for i in range(5)))
