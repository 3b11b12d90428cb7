"""One full configs[3] run: the reference's training schedule for inb_lan.yaml — 6 epochs x 500 iterations
(train_net.py:131-158 around Trainer.train; configs/inb/inb_lan.yaml over inb_377.yaml: smpl_thresh 0.1, lr 1e-3, eps 1e-15,
pair_loss_weight 1e-4, 64 x 64 patches x 64 samples, exponential LR decay per epoch) — through invr.driver.train on one MI355X.

There is no dataset here (MonoCap is licensed) and no VGG weights (the LPIPS branch needs them), so
  * the target is SYNTHETIC and multi-frame: a "teacher" network of the same architecture (N(0, 0.1^2) tables, seed 7) renders F poses
    of the synthetic body at 512 x 512; the student (default initialisation, another seed) is trained on random 64 x 64 patches of
    those frames,
  * the image term is the plain MSE (use_lpips False), the regularisers are the reference's.
Reports: wall clock for the 3000 iterations, loss / patch PSNR per 100 iterations, whole-image PSNR (evaluators/if_nerf.py:28-31 form)
of every frame before and after, and the first 20 losses of a fixed-noise prefix against the CPU oracle's training loop
(tests/oracle_train.py + torch.optim.Adam).  Writes one JSON (argv[1], default profiles/r4_configs3_full_run.json)."""
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                              # noqa: E402
import torch                                    # noqa: E402
import invr                                     # noqa: E402,F401
from invr import scene as scene_mod, driver     # noqa: E402
from invr.config import make_cfg                # noqa: E402
from invr.network import Network                # noqa: E402
from invr.trainer import NetworkWrapper         # noqa: E402
import bench                                    # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r4_configs3_full_run.json')
F = int(os.environ.get('LAN_FRAMES', '8'))
EPOCHS, EP_ITER = int(os.environ.get('LAN_EPOCHS', '6')), int(os.environ.get('LAN_EP_ITER', '500'))
ORACLE_STEPS = int(os.environ.get('LAN_ORACLE_STEPS', '20'))
PATCHES_PER_FRAME = 16
RES, SIDE, S, LR = 512, 64, 64, 1e-3
dev = torch.device('cuda', 0)
cfg = make_cfg(N_samples=S, smpl_thresh=0.1, pair_loss_weight=1e-4)          # inb_lan.yaml over inb_377.yaml

# ---- the synthetic multi-frame target ---------------------------------------------------------------------------------------------
teacher = bench.build_model(copy.deepcopy(cfg), dev, seed=7)
frames_cpu, frames = [], []
for f in range(F):
    bnp, _ = scene_mod.make_scene(RES, RES, seed=0, frame=(3 + 11 * f) % 100, cam_dist=1.8, pose_seed=f)
    b = scene_mod.to_torch(bnp)
    gb = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        o = teacher.render_rays(gb, gb['ray_o'][0], gb['ray_d'][0], gb['near'][0], gb['far'][0], S, want_raw=False)
    gb['rgb'] = o['rgb_map'][None].clone()
    b['rgb'] = gb['rgb'].cpu()
    frames_cpu.append(b)
    frames.append(gb)
teacher._ws = None
del teacher
torch.cuda.empty_cache()


def make_patch(b, y0, x0):
    """the rays of a 64 x 64 window of frame batch b (host tensors), as scene.make_scene(crop=...) lays a patch out"""
    H = W = RES
    mask = b['mask_at_box'][0].reshape(H, W)
    win = torch.zeros(H, W, dtype=torch.bool)
    win[y0:y0 + SIDE, x0:x0 + SIDE] = True
    keep = win.reshape(-1)[mask.reshape(-1)]
    p = dict(b)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        p[k] = b[k][:, keep]
    p['mask_at_box'] = (mask & win).reshape(1, -1)
    return p


rng = np.random.RandomState(5)
pool_cpu = []
for b in frames_cpu:
    pix = torch.nonzero(b['mask_at_box'][0].reshape(RES, RES))
    lit = pix[(b['rgb'][0].sum(1) > 0.05)]                                     # windows centred on pixels the teacher's body covers
    for _ in range(PATCHES_PER_FRAME):
        c = lit[rng.randint(len(lit))]
        y0, x0 = int(min(max(int(c[0]) - SIDE // 2, 0), RES - SIDE)), int(min(max(int(c[1]) - SIDE // 2, 0), RES - SIDE))
        pool_cpu.append(make_patch(b, y0, x0))
pool = [{k: v.to(dev) for k, v in p.items()} for p in pool_cpu]
order = rng.randint(len(pool), size=EPOCHS * EP_ITER)


def student(seed):
    torch.manual_seed(seed)
    with torch.device(dev):
        net = Network(cfg=copy.deepcopy(cfg))
    return net.to(dev).train()


def whole_image_psnr(net):
    net.eval()
    vals = []
    with torch.no_grad():
        for gb in frames:
            o = net.render_rays(gb, gb['ray_o'][0], gb['ray_d'][0], gb['near'][0], gb['far'][0], S, want_raw=False)
            H = W = RES
            mask = gb['mask_at_box'][0].reshape(H, W).cpu().numpy()
            pred, gt = np.zeros((H, W, 3)), np.zeros((H, W, 3))
            pred[mask] = o['rgb_map'].cpu().numpy()
            gt[mask] = gb['rgb'][0].cpu().numpy()
            vals.append(float(driver.psnr_metric(pred, gt)))
    net._ws = None
    net.train()
    return vals


result = {'config': 'configs[3]: inb_lan.yaml over inb_377.yaml (smpl_thresh 0.1, lr 1e-3, eps 1e-15, pair_loss_weight 1e-4), 64x64 patches x 64 samples, '
                    '%d epochs x %d iterations, full-size model (1.09 GB tables), MSE image term (no VGG weights here)' % (EPOCHS, EP_ITER),
          'target': 'synthetic: %d poses of the synthetic body rendered at %dx%d by a teacher network of the same architecture; %d patches per frame'
                    % (F, RES, RES, PATCHES_PER_FRAME)}

# ---- (1) the first iterations against the CPU oracle's training loop (fixed jitter / pair noise) -------------------------------------
if ORACLE_STEPS:
    from tests import oracle_train as OT
    net = student(1)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    ks = [int(order[i]) for i in range(ORACLE_STEPS)]
    jit = [torch.rand(pool_cpu[k]['ray_o'].shape[1], S, generator=g) for k in ks]
    noi = [torch.rand(pool_cpu[k]['ray_o'].shape[1] * S * 5, 3, generator=g) for k in ks]
    wrap = NetworkWrapper(net)
    opt = driver.make_optimizer(net, lr=LR, eps=1e-15)
    cur = {'i': 0}
    wrap.renderer._jitter = lambda shape, device: jit[cur['i']].to(device)
    wrap.renderer._pair_noise_dense = lambda rows, device: noi[cur['i']].to(device)[:rows]
    mine = []
    for i in range(ORACLE_STEPS):
        cur['i'] = i
        loss, _ = driver.train_step(wrap, opt, dict(pool[ks[i]]), i + 1)
        mine.append(loss)
    torch.cuda.synchronize()
    mine = [float(x) for x in mine]
    del wrap, opt, net
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    sd = {k: v.clone() for k, v in sd0.items()}
    train_keys = [k for k, p in Network(cfg=copy.deepcopy(cfg)).named_parameters() if p.requires_grad]
    for k in train_keys:
        sd[k].requires_grad_()
    ref_opt = torch.optim.Adam([{'params': [sd[k]], 'lr': LR} for k in train_keys], LR, eps=1e-15)
    ref = []
    for i in range(ORACLE_STEPS):
        if i + 1 == 1:
            OT.adopt_batch_bounds(sd, cfg, pool_cpu[ks[i]])
        loss, _ = OT.train_loss(sd, cfg, pool_cpu[ks[i]], jit[i], noi[i], chunk=1024)
        ref_opt.zero_grad(set_to_none=True)
        loss.backward()
        ref_opt.step()
        ref.append(float(loss))
    result['first_losses'] = {'hip': mine, 'cpu_oracle': ref, 'max_rel_diff': float(np.max(np.abs(np.array(mine) - np.array(ref)) / np.abs(np.array(ref)))),
                              'cpu_oracle_seconds': time.perf_counter() - t0}
    del sd, ref_opt
    print('first %d losses: max rel diff vs the CPU oracle %.2e' % (ORACLE_STEPS, result['first_losses']['max_rel_diff']), flush=True)

# ---- (2) the schedule ---------------------------------------------------------------------------------------------------------------------
net = student(1)
result['psnr_before'] = whole_image_psnr(net)
wrap = NetworkWrapper(net)
opt = driver.make_optimizer(net, lr=LR, eps=1e-15)
sched = driver.ExponentialLR(opt, decay_epochs=1000, gamma=0.1)
psnrs = []
it = [0]


def batch_fn(epoch, index):
    k = int(order[it[0] % len(order)])
    it[0] += 1
    return dict(pool[k])


out = driver.train(wrap, opt, batch_fn, EPOCHS, EP_ITER, scheduler=sched, on_step=lambda e, i, l, s: psnrs.append(s['psnr']))
torch.cuda.synchronize()
losses = np.array(out['losses'])
ps = torch.cat([p.reshape(1) for p in psnrs]).cpu().numpy()
n100 = len(losses) // 100
result.update({
    'iterations': out['iterations'], 'seconds': out['seconds'], 'ms_per_iteration': out['seconds'] / max(1, out['iterations']) * 1e3,
    'ray_samples_per_sec': out['ray_samples'] / out['seconds'],
    'loss_per_100_iterations': [float(losses[i * 100:(i + 1) * 100].mean()) for i in range(n100)],
    'patch_psnr_per_100_iterations': [float(ps[i * 100:(i + 1) * 100].mean()) for i in range(n100)],
    'first_loss': float(losses[0]), 'last_100_loss': float(losses[-100:].mean()),
    'lr_end': float(opt.param_groups[0]['lr']),
})
result['psnr_after'] = whole_image_psnr(net)
result['psnr_gain_db'] = float(np.mean(result['psnr_after']) - np.mean(result['psnr_before']))
print(json.dumps({k: result[k] for k in ('iterations', 'seconds', 'ms_per_iteration', 'first_loss', 'last_100_loss', 'psnr_gain_db')}), flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(result, open(out_path, 'w'), indent=1)
