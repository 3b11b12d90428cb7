"""Golden vectors for two non-default switches of the reference's hot path, from the IMPORTED reference (run in the build container,
where /root/reference exists; the same stand-ins as make_golden.py):
  * cfg.aggr = 'mean'   (inb_part_network_multiassign.py:236-239: raw / occ = the mean over the five parts, zeros for unflagged parts)
  * cfg.random_bg = True (inb_renderer.py:72 passes the flag to volume_rendering as render_weights' epsilon, net_utils.py:12-18)
Same scene (64 x 64, seed 0), same parameters (seed 7), same 32 samples as inb377_small.npz.  Stores, per mode: the eval render's
rgb_map / acc_map / non-zero raw rows, and a train-mode forward on the 256 train rays of inb377_small.npz (fixed jitter / pair noise)
with its loss and every small parameter gradient (the part tables: the touched rows' first column).  -> tests/golden/modes_small.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG      # noqa: E402


MODES = (('mean', 'aggr', 'mean'), ('rbg', 'random_bg', True))
OUT = 'modes_small.npz'
if os.environ.get('GOLDEN_MODES') == 'dist':          # round 5: the two distance-weighted merges (:240-251) -> modes_dist_small.npz
    MODES = (('dist', 'aggr', 'dist'), ('mind', 'aggr', 'mindist'))
    OUT = 'modes_dist_small.npz'


def main():
    import invr  # noqa: F401
    from invr import scene, params
    from invr.config import make_cfg
    S = 32
    rcfg = MG.import_reference(S)
    torch.manual_seed(0)
    from lib.networks.make_network import make_network
    from lib.networks.renderer.make_renderer import make_renderer
    cfg = make_cfg(table_log2=MG.TABLE_LOG2, N_samples=S)
    net = make_network(rcfg)
    sd = params.init_state_dict(cfg, seed=7)
    net.load_state_dict(sd, strict=True)
    batch_np, _ = scene.make_scene(64, 64, seed=0)
    batch = scene.to_torch(batch_np)
    renderer = make_renderer(rcfg, net)
    base = np.load(os.path.join(HERE, 'inb377_small.npz'))
    tsel = base['train_rays'].astype(np.int64)
    jit = torch.from_numpy(base['train_jitter'])
    out = {}
    rcfg.defrost()
    for tag, key, val in MODES:
        old = rcfg[key]
        rcfg[key] = val
        try:
            net.eval()
            with torch.no_grad():
                ret = renderer.render(dict(batch))
            out[tag + '_rgb_map'] = MG.tnp(ret['rgb_map'])
            out[tag + '_acc_map'] = MG.tnp(ret['acc_map'])
            raw = MG.tnp(ret['raw'])[0]
            nz = np.nonzero(np.abs(raw).sum(1) != 0)[0].astype(np.int32)
            out[tag + '_raw_nz_idx'] = nz
            out[tag + '_raw_nz'] = raw[nz]
            # train-mode forward + gradients on the golden's train rays
            net.train()
            net.zero_grad(set_to_none=True)
            tb_ = dict(batch)
            for k in ['ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy']:
                tb_[k] = batch[k][:, tsel]
            pair = {}
            _rand, _rand_like = torch.rand, torch.rand_like

            def rand(*a, **k):
                return jit.clone()

            def rand_like(x, **k):
                pair['u'] = _rand(x.shape, generator=torch.Generator().manual_seed(22))
                return pair['u'].clone()
            torch.rand, torch.rand_like = rand, rand_like
            try:
                tret = renderer.render(dict(tb_))
            finally:
                torch.rand, torch.rand_like = _rand, _rand_like
            out[tag + '_train_pair_u'] = MG.tnp(pair['u']) if 'u' in pair else np.zeros((1, 0, 3), np.float32)
            out[tag + '_train_rgb_map'] = MG.tnp(tret['rgb_map'])
            out[tag + '_train_tocc'] = MG.tnp(tret['tocc'])
            loss = ((tret['rgb_map'] - tb_['rgb']) ** 2).mean() + 0.1 * tret['reg_distortion_loss'].mean() \
                + 0.1 * torch.norm(tret['resd'], dim=2).mean()
            loss.backward()
            out[tag + '_train_loss'] = MG.tnp(loss)
            for k, p in net.named_parameters():
                if p.grad is None:
                    continue
                g = MG.tnp(p.grad)
                if g.size > 4096:
                    flat = g.reshape(-1, g.shape[-1])
                    rows = np.nonzero(np.abs(flat).sum(1) > 0)[0]
                    out['%s_grad_rows::%s' % (tag, k)] = rows.astype(np.int32)
                    out['%s_grad_vals::%s' % (tag, k)] = flat[rows][:, :1].copy()
                else:
                    out['%s_grad::%s' % (tag, k)] = g
        finally:
            rcfg[key] = old
    path = os.path.join(HERE, OUT)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) / 1e6, 'MB;', len(out), 'arrays')


if __name__ == '__main__':
    main()
