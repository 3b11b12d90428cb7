"""CPU model of VERDICT r4 #1's proposal for k_knn_pairs: an exact per-(lattice cell, part) candidate vertex list
{v : |c - v| <= d4(c) + 2h} (c = cell centre, h = half diagonal, d4 = exact 4th-nearest distance from c: the 4-NN of every point of
the cell lie in it, by the 1-Lipschitz bound) scanned per LANE instead of the wave-uniform pruned cluster sweep.

Step 1 (wave machine, ~30 s at 256x256): the kernel sources on the CPU give the survivors of the bench frame in the kernel's own
(depth-windowed) order.  Step 2 (scipy KD-tree): list lengths per cell and, per ticket of 64 consecutive survivors, the LONGEST list
among its scanning lanes — a wave walks its lanes' lists in lock-step, so that maximum is the trip count of the per-lane loop.
Usage: python tools/knn_list_model.py [RES]        (default 256; prints the table of profiles/r5_knn_list_model.md)"""
import os
import sys
import time

import numpy as np
import torch
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim import harness          # noqa: E402


def survivors(RES, S=128):
    with harness.activate():
        from invr import _abi, scene, stages
        from invr.config import make_cfg
        from invr.network import Network
        cfg = make_cfg(table_log2=12, N_samples=S)
        net = Network(cfg=cfg).eval()
        bnp, _ = scene.make_scene(RES, RES, seed=0, cam_dist=1.8)
        gb = scene.to_torch(bnp)
        ctx = net.prepare(gb)
        ro, rd, nr, fa = (gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))
        t = time.time()
        out = net.geometry_pass(ctx, ro, rd, nr, fa, S)
        print('geometry pass on the wave machine: %.1f s' % (time.time() - t))
        Na = int(out['stats'].numpy()[0])
        v = _abi.ws_views(*out['_ws'])
        act = v['active_idx'][:Na].clone()
        pts, _ = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, act, want_dirs=False)
        return pts.numpy().astype(np.float64), bnp


def main():
    RES = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    pts, b = survivors(RES)
    pp, l2, pb = b['part_pts'][0], b['lengths2'].reshape(-1), b['pbounds'].reshape(2, 3)
    dims = np.array(b['pbw'].shape[1:4])
    Na, thresh = len(pts), 0.05
    csz = (pb[1] - pb[0]) / (dims - 1)
    h = 0.5 * np.linalg.norm(csz)
    ci = np.clip(np.floor((pts - pb[0]) / csz).astype(np.int64), 0, dims - 2)
    cell = (ci[:, 0] * dims[1] + ci[:, 1]) * dims[2] + ci[:, 2]
    ucell, inv = np.unique(cell, return_inverse=True)
    cen = pb[0] + (np.stack([ucell // (dims[2] * dims[1]), (ucell // dims[2]) % dims[1], ucell % dims[2]], 1) + 0.5) * csz
    n_t = Na // 64
    print('%d survivors, %d tickets, %d cells hold survivors, cell %.1f cm (half diagonal %.2f cm)' % (Na, n_t, len(ucell), 100 * csz[0], 100 * h))
    print('part  vertices  near-type cells  list mean / p90 / max (vertices)   tickets with a near lane   mean of the LONGEST list per such ticket   distinct cells per such ticket')
    for p in range(5):
        tree = cKDTree(pp[p, :l2[p]].astype(np.float64))
        d4 = tree.query(cen, k=4)[0]
        cnt = np.array([len(x) for x in tree.query_ball_point(cen, d4[:, 3] + 2 * h)])
        d1p = tree.query(pts, k=1)[0]
        near = d1p < 1.01 * thresh                         # lanes whose exact 4-NN the kernel needs from a NEAR part (band lanes: 0.47-0.68 m)
        lst = cnt[inv]
        L, SC = lst[:n_t * 64].reshape(n_t, 64), near[:n_t * 64].reshape(n_t, 64)
        has = SC.any(1)
        mx = np.where(SC, L, 0).max(1)
        ncell = np.array([len(np.unique(cell[t * 64:(t + 1) * 64][SC[t]])) for t in np.nonzero(has)[0][::5]])
        near_cells = d4[:, 0] - h < 1.01 * thresh
        print('%4d  %8d  %15d  %5.1f / %3.0f / %3d %28d %35.1f %36.1f'
              % (p, l2[p], near_cells.sum(), lst[near].mean(), np.percentile(lst[near], 90), lst[near].max(), has.sum(), mx[has].mean(), ncell.mean()))


if __name__ == '__main__':
    main()
