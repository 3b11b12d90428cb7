"""Helper process of the survivor-order tests: renders one seeded frame and saves rgb_map / acc_map / raw / occ + the statistics.
The order (INVR_ORDER, csrc/invr_abi.hip) is fixed when the library first renders, hence one process per order.
usage: python tests/order_frame.py <out.npz> <device> <res> <samples> [hostsim]"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, dev, res, S = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4].split(',')[0])
    more = [int(x) for x in sys.argv[4].split(',')[1:]]              # further sample counts, rendered on a 1/4 ray subset: keys '<name>_<S>'
    ctx = contextlib.nullcontext()
    if len(sys.argv) > 5 and sys.argv[5] == 'hostsim':
        from tests.hostsim import harness
        ctx = harness.activate()
    with ctx:
        import invr  # noqa: F401
        from invr import scene
        from invr.config import make_cfg
        from invr.network import Network
        torch.manual_seed(77)
        cfg = make_cfg(table_log2=12, N_samples=S)
        net = Network(cfg=cfg)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for name, p in net.named_parameters():
                if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        net = net.to(dev).eval()
        bnp, _ = scene.make_scene(res, res, seed=1, pose_scale=1.0, frame=17, cam_dist=1.8)
        gb = {k: v.to(dev) for k, v in scene.to_torch(bnp).items()}
        junk = torch.full((int(res * res * S * 6),), float('nan'), device=dev)          # the outputs come from dirtied allocator blocks
        del junk
        o = net.render_rays(gb, gb['ray_o'][0], gb['ray_d'][0], gb['near'][0], gb['far'][0], S, want_raw=True)
        v = None
        from invr import _abi
        v = _abi.ws_views(*o['_ws'])
        na = int(o['stats'][0])
        res_d = dict(rgb=o['rgb_map'].cpu().numpy(), acc=o['acc_map'].cpu().numpy(), raw=o['raw'].cpu().numpy(), occ=o['occ'].cpu().numpy(),
                     stats=o['stats'].cpu().numpy(), act=v['active_idx'][:na].clone().cpu().numpy())
        sub = torch.arange(0, gb['ray_o'].shape[1], 4, device=dev)
        for S2 in more:
            o = net.render_rays(gb, gb['ray_o'][0][sub], gb['ray_d'][0][sub], gb['near'][0][sub], gb['far'][0][sub], S2, want_raw=True)
            v = _abi.ws_views(*o['_ws'])
            na = int(o['stats'][0])
            res_d.update({'rgb_%d' % S2: o['rgb_map'].cpu().numpy(), 'occ_%d' % S2: o['occ'].cpu().numpy(),      # (raw = 16 B per ray-sample: left out)
                          'stats_%d' % S2: o['stats'].cpu().numpy(), 'act_%d' % S2: v['active_idx'][:na].clone().cpu().numpy()})
        np.savez(out, **res_d)


if __name__ == '__main__':
    main()
