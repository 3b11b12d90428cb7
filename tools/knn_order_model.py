"""CPU model of the KNN sweep's work under different ORDERS of the survivors (no GPU needed).

k_knn_pairs hands 64 consecutive survivors to a wave and prunes clusters / sub-clusters per WAVE (a sub-cluster of 16 vertices is
scanned if any lane needs it), so its work depends on how compact in space 64 consecutive survivors are.  The compaction's order is
ray-major (64 consecutive survivors = the front and back depth segments of ~7 neighbouring rays, mean bounding-box diagonal 24 cm).
This tool runs the kernel SOURCE on the CPU wave machine (tests/hostsim, built with -DKNN_PROF: the kernel's own phase counters)
with the survivors permuted into candidate orders and reports, per order: part scans, candidate clusters visited and 16-vertex
sub-clusters scanned per frame — the quantities the sweep (65 % of the kernel's time, profiles/r3_knn_phases.md) and the
classification (17 %) scale with.  Orders:
  ray-major            what the compaction produces today
  tile R x window D    survivors grouped by tiles of R consecutive rays, inside a tile by depth windows of D samples, inside a
                       window ray-major — what a compaction with a different word order could produce without any sort
  cell-sorted          by 4 cm lattice cell (round 3's experiment: -20 % kernel time, but the sort cost more than it saved)
  ... inside blocks    the same inside blocks of C consecutive survivors only: what one workgroup per block could do in LDS
  Morton               Morton order of 2 cm cells, inside blocks / over the frame
Usage: python tools/knn_order_model.py [RES] [S]      (default 256 128; 512 128 takes ~15 min)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim import harness          # noqa: E402


def main():
    RES = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    with harness.activate(extra_flags=('-DKNN_PROF', '-DHOSTSIM_KNN_PERM')) as cnt:
        from invr import _abi, scene, stages
        from invr.config import make_cfg
        from invr.network import Network
        L = cnt.lib
        L.invr_debug_knn_prof.argtypes = [C.c_void_p, C.c_int]
        L.hostsim_set_knn_perm.argtypes = [C.c_void_p]
        cfg = make_cfg(table_log2=12, N_samples=S)
        net = Network(cfg=cfg).eval()
        bnp, _ = scene.make_scene(RES, RES, seed=0, cam_dist=1.8)
        gb = scene.to_torch(bnp)
        ctx = net.prepare(gb)
        ro, rd, nr, fa = (gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))

        def run(perm):
            prof = (C.c_ulonglong * 32)()
            L.invr_debug_knn_prof(None, 1)
            keep = None
            if perm is not None:
                keep = torch.from_numpy(np.ascontiguousarray(perm.astype(np.int32)))
            L.hostsim_set_knn_perm(C.c_void_p(keep.data_ptr()) if keep is not None else None)
            t = time.time()
            out = net.geometry_pass(ctx, ro, rd, nr, fa, S)
            dt = time.time() - t
            L.invr_debug_knn_prof(prof, 0)
            L.hostsim_set_knn_perm(None)
            return out, list(prof), dt

        out, base, dt = run(None)
        st = out['stats'].numpy()
        Na = int(st[0])
        v = _abi.ws_views(*out['_ws'])
        act = v['active_idx'][:Na].clone().numpy().astype(np.int64)
        pts, _ = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, torch.from_numpy(act.astype(np.int32)), want_dirs=False)
        pts = pts.numpy()
        ray, smp = act // S, act % S
        print('frame %dx%dx%d: %d rays, %d survivors (%.1f s per geometry pass on the wave machine)' % (RES, RES, S, ro.shape[0], Na, dt))

        def report(name, prof, perm):
            order = perm if perm is not None else np.arange(Na)
            n = Na // 64 * 64
            p = pts[order[:n]].reshape(-1, 64, 3)
            diag = np.linalg.norm(p.max(1) - p.min(1), axis=1).mean()
            print('%-34s tickets %6d  part scans %7d  clusters visited %8d  sub-clusters scanned %8d (%.3f of ray-major; %d of them in part scans without a near lane); insert branch taken for %d of %d pair records  mean ticket diagonal %.1f cm'
                  % (name, prof[8], prof[9], prof[10], prof[11], prof[11] / max(base[11], 1), prof[7], prof[12], 8 * prof[11], 100 * diag))

        report('ray-major (today)', base, None)
        for R, D in [tuple(int(v) for v in a.split('x')) for a in os.environ.get('KNN_ORDERS', '64x16,64x32,16x16,256x16,64x8').split(',')]:
            key = (ray // R) * (S // D + 1) * R * S + (smp // D) * R * S + (ray % R) * S + smp
            perm = np.argsort(key, kind='stable')
            _, prof, _ = run(perm)
            report('tile %d rays x window %d samples' % (R, D), prof, perm)
        # 2-D pixel tiles (round 6): rays grouped by TW x TH pixel tiles of the image (the rays of a frame are the True pixels of
        # batch['mask_at_box'] in row-major order), inside a tile by depth windows of D samples, inside a window pixel-major
        if os.environ.get('KNN_TILES2D'):
            mask = np.asarray(bnp['mask_at_box']).reshape(RES, RES)
            py, px = np.nonzero(mask)
            assert py.shape[0] == ro.shape[0]
            for TW, TH, D in [tuple(int(v) for v in a.split('x')) for a in os.environ['KNN_TILES2D'].split(',')]:
                ty, tx = py[ray] // TH, px[ray] // TW
                inner = (py[ray] % TH) * TW + (px[ray] % TW)
                key = ((ty * 4096 + tx) * (S // D + 1) + smp // D) * (TW * TH * S) + inner * S + smp
                perm = np.argsort(key, kind='stable')
                _, prof, _ = run(perm)
                report('2-D tile %dx%d px x window %d' % (TW, TH, D), prof, perm)
        if os.environ.get('KNN_ORDERS'):
            return
        lo = pts.min(0)
        q = np.floor((pts - lo) / 0.04).astype(np.int64)
        key = (q[:, 0] * 4096 + q[:, 1]) * 4096 + q[:, 2]
        perm = np.argsort(key, kind='stable')
        _, prof, _ = run(perm)
        report('cell-sorted (4 cm cells)', prof, perm)
        # what ONE workgroup per block of C consecutive survivors could do in LDS (no global atomics): sort the block by cell
        for Cb, cell in ((4096, 0.04), (4096, 0.02), (1024, 0.04), (16384, 0.04)):
            q = np.floor((pts - lo) / cell).astype(np.int64)
            key = (np.arange(Na) // Cb) * (1 << 40) + (q[:, 0] * 1024 + q[:, 1]) * 1024 + q[:, 2]
            perm = np.argsort(key, kind='stable')
            _, prof, _ = run(perm)
            report('cell-sorted inside blocks of %d (%.0f cm)' % (Cb, cell * 100), prof, perm)
        # Morton order of the cells instead of x-major
        def spread(x):
            x = x & 0x3ff
            x = (x | (x << 16)) & 0x30000ff
            x = (x | (x << 8)) & 0x300f00f
            x = (x | (x << 4)) & 0x30c30c3
            return (x | (x << 2)) & 0x9249249
        q = np.floor((pts - lo) / 0.02).astype(np.int64)
        mort = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        for Cb in (4096, 1 << 40):
            key = (np.arange(Na) // Cb) * (1 << 40) + mort
            perm = np.argsort(key, kind='stable')
            _, prof, _ = run(perm)
            report('Morton (2 cm) inside blocks of %s' % ('%d' % Cb if Cb < 1 << 30 else 'the frame'), prof, perm)


if __name__ == '__main__':
    main()
