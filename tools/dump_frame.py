"""Render the bench frame at the four test poses and save rgb_map / acc_map / raw / occ (+ SHA-256 of each) under --out.
Used for bit-equality A/B of two source trees on one GPU box:  python tools/dump_frame.py --root <tree> --out <dir>;
python tools/dump_frame.py --compare <dirA> <dirB>."""
import argparse
import copy
import hashlib
import os
import sys

import torch

POSES = [dict(seed=0, pose_scale=0.5, frame=3, cam_dist=1.8, thresh=0.05),
         dict(seed=1, pose_scale=1.0, frame=17, cam_dist=1.8, thresh=0.05),
         dict(seed=2, pose_scale=1.2, frame=60, cam_dist=2.2, thresh=0.1),
         dict(seed=3, pose_scale=0.8, frame=99, cam_dist=2.6, thresh=0.05)]


def dump(root, out, S, full_rows):
    sys.path.insert(0, root)
    import invr  # noqa: F401
    from invr import scene
    from invr.config import make_cfg
    from invr.network import Network
    os.makedirs(out, exist_ok=True)
    dev = 'cuda:0'
    cfg0 = make_cfg(N_samples=S)
    torch.manual_seed(1234)                     # the MLP weights take torch's default init from the global generator
    with torch.device(dev):
        net = Network(cfg=cfg0)
    net = net.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
                p.normal_(0.0, 0.1, generator=g)
            elif p.dim() == 1 and p.is_floating_point():
                p.normal_(0.0, 0.05, generator=g)       # non-zero biases / latents: every term of the MLPs is exercised
    if full_rows:
        cfg0.eval_row_sums = False
    for k, kw in enumerate(POSES):
        kw = dict(kw)
        cfg = copy.deepcopy(cfg0)
        cfg.smpl_thresh = kw.pop('thresh')
        net.cfg = cfg
        bnp, _ = scene.make_scene(512, 512, **kw)
        gb = {k_: v.to(dev) for k_, v in scene.to_torch(bnp).items()}
        ctx = net.prepare(gb)
        ro, rd, nr, fa = (gb[k_][0] for k_ in ('ray_o', 'ray_d', 'near', 'far'))
        o = net.render_rays(ctx, ro, rd, nr, fa, S, want_raw=True)
        torch.cuda.synchronize()
        rec = {}
        for name in ('rgb_map', 'acc_map', 'raw', 'occ'):
            t = o[name].detach().cpu().contiguous()
            rec[name] = hashlib.sha256(t.numpy().tobytes()).hexdigest()
            torch.save(t, os.path.join(out, 'pose%d_%s.pt' % (k, name)))
        print('pose', k, 'stats', o['stats'].cpu().tolist()[:7], {n: h[:12] for n, h in rec.items()}, flush=True)


def compare(a, b):
    bad = 0
    for f in sorted(os.listdir(a)):
        if not f.endswith('.pt'):
            continue
        x, y = torch.load(os.path.join(a, f)), torch.load(os.path.join(b, f))
        eq = x.shape == y.shape and torch.equal(x, y)
        d = float((x - y).abs().max()) if x.shape == y.shape else float('nan')
        print('%-22s %s  max|d| = %.3e' % (f, 'EQUAL' if eq else 'DIFFERENT', d))
        bad += 0 if eq else 1
    n = len([f for f in os.listdir(a) if f.endswith('.pt')])
    print('A/B over %d tensors:' % n, 'bit-identical' if bad == 0 and n else '%d tensors differ' % bad)
    return bad


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap.add_argument('--out')
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--full-rows', action='store_true')
    ap.add_argument('--compare', nargs=2)
    a = ap.parse_args()
    if a.compare:
        sys.exit(1 if compare(*a.compare) else 0)
    dump(os.path.abspath(a.root), a.out, a.samples, a.full_rows)
