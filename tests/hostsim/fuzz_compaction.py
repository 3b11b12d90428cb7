"""TEST INFRASTRUCTURE — random frames (ray subsets in random order, power-of-two sample counts 8..512, random scenes) through the host
build of the kernels: the (ray-sample -> slot) map of the compaction against the active list, the windowed key order, exact / too small
survivor capacities, ray-permutation invariance.  python tests/hostsim/fuzz_compaction.py [seed]   (40 cases, ~1 min; run by hand)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.hostsim import harness
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
with harness.activate() as cnt:
    from invr import _abi, scene
    from invr.config import make_cfg
    from invr.network import Network
    torch.manual_seed(1)
    cfg = make_cfg(table_log2=12, N_samples=64)
    net = Network(cfg=cfg).eval()
    tab = torch.tensor([bin(x).count('1') for x in range(256)])
    bad = 0
    for it in range(40):
        res = int(rng.choice([24, 40, 56, 72]))
        S = int(rng.choice([8, 16, 32, 64, 128, 256, 512]))
        bnp, _ = scene.make_scene(res, res, seed=int(rng.integers(0, 5)), frame=int(rng.integers(0, 100)), cam_dist=float(rng.uniform(1.5, 2.6)), pose_scale=float(rng.uniform(0.3, 1.2)))
        gb = scene.to_torch(bnp)
        n = gb['ray_o'].shape[1]
        k = int(rng.integers(1, n + 1))
        sel = torch.from_numpy(rng.permutation(n)[:k].astype(np.int64))
        ro, rd, nr, fa = (gb[q][0][sel] for q in ('ray_o', 'ray_d', 'near', 'far'))
        ctx = net.prepare(gb)
        full = net.render_rays(ctx, ro, rd, nr, fa, S, want_raw=True)
        na = int(full['stats'][0])
        v = _abi.ws_views(*full['_ws'])
        act = v['active_idx'][:na].long().clone()
        # slot map
        if na:
            word, bit = v['mask'][act >> 6], act & 63
            asc = act.sort()[0]
            below = word & ((torch.ones_like(bit) << bit) - 1)
            slot_w = v['word_off'][act >> 6].long() + sum(tab[(below >> (8 * j)) & 255] for j in range(8))
            slot_b = v['byte_off'][act >> 3].long() + tab[((word >> (bit & ~7)) & 255) & ((torch.ones_like(bit) << (bit & 7)) - 1)]
            rows = 8192 // S
            ray, smp = act // S, act % S
            key = ((ray // rows) * (S // 8) + smp // 8) * 8192 + (ray % rows) * 8 + smp % 8
            assert bool((key[1:] > key[:-1]).all()), ('key order', it)
            assert torch.equal(slot_b, torch.arange(na)), ('slot map', it, res, S, k)
            assert bool((asc[1:] > asc[:-1]).all())
            # survivors = nonzero occ... (raw[:,3] nonzero subset of survivors)
            nz = (full['raw'][:, 3] != 0).nonzero()[:, 0]
            assert bool(torch.isin(nz, asc).all())
        rgb, raw = full['rgb_map'].clone(), full['raw'].clone()
        # capacity exactly na, and too small
        if na > 2:
            o = net.render_rays(ctx, ro, rd, nr, fa, S, want_raw=True, max_active=na)
            assert int(o['stats'][6]) == 0 and torch.equal(o['rgb_map'], rgb) and torch.equal(o['raw'], raw), ('cap=na', it)
            o = net.render_rays(ctx, ro, rd, nr, fa, S, max_active=int(rng.integers(1, na)))
            assert int(o['stats'][6]) == 1, ('overflow flag', it)
        # permutation invariance
        p = torch.from_numpy(rng.permutation(k).astype(np.int64))
        o = net.render_rays(ctx, ro[p], rd[p], nr[p], fa[p], S)
        assert torch.equal(o['rgb_map'], rgb[p]), ('perm', it, res, S, k)
        print(it, res, S, k, na, 'ok', flush=True)
    print('anomalies', cnt.anomalies)
