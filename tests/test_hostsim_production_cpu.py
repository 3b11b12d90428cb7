"""CPU: the PRODUCTION kernels of the eval frame (k_knn_pairs, k_warp_pairs + k_deform_pairs_slice, k_part_encode_rs_xcd, k_part_occ_all /
k_winner_lists / k_part_rgb_all) pinned directly — the test bodies of tests/test_gpu_production_kernels.py on a reduced frame
(96 x 96 pixels x 64 samples, 2^12-row tables) through the host build of the kernel sources (tests/hostsim).  The frame is large
enough for several 4096-slot groups (segmented pair / winner lists, the colour kernel's segment cursor) and for every class of the
KNN's lattice cells; the whole-frame sizes stay with -m gpu."""
import copy
import os

import pytest
import torch

import tests.test_gpu_production_kernels as P
from tests.hostsim import harness

SMALL = dict(RES=96, S=64,
             MIN=dict(na=16000, listed=16000, oracle_subset=1500, oracle_chunk=500, enc_take=4000, enc_total=9000, enc_inside=2000,
                      strict_rays=48, occ=40))
if os.environ.get('HOSTSIM_FRAME'):        # e.g. HOSTSIM_FRAME=256,128: a one-off larger frame (> 64 slot groups: the segment cursor's windows)
    SMALL['RES'], SMALL['S'] = (int(x) for x in os.environ['HOSTSIM_FRAME'].split(','))
    _f = SMALL['RES'] ** 2 * SMALL['S'] / (96 * 96 * 64)
    SMALL['MIN'].update(na=int(16000 * _f), listed=int(16000 * _f))
POSE_IDS = [int(x) for x in os.environ.get('HOSTSIM_POSES', '2').split(',')]      # default: pose_scale 1.2 at the inb_lan threshold 0.1 (HOSTSIM_POSES=0,1,2,3: all)


@pytest.fixture(scope='module', autouse=True)
def hostsim():
    old = {k: getattr(P, k) for k in ('DEV', 'RES', 'S', 'MIN')}
    P.DEV = 'cpu'
    P.RES, P.S, P.MIN = SMALL['RES'], SMALL['S'], SMALL['MIN']
    try:
        with harness.activate() as counters:
            yield counters
            assert counters.anomalies == 0, counters.anomalies
    finally:
        for k, v in old.items():
            setattr(P, k, v)


@pytest.fixture(scope='module')
def small_net(hostsim):
    from invr.config import make_cfg
    from invr.network import Network
    torch.manual_seed(11)
    cfg = make_cfg(table_log2=12, N_samples=SMALL['S'])
    net = Network(cfg=copy.deepcopy(cfg)).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
                p.normal_(0.0, 0.1, generator=g)
    return cfg, net


@pytest.fixture(scope='module', params=POSE_IDS, ids=['pose%d' % i for i in POSE_IDS])
def fr(request, small_net):
    return P.make_frame(request.param, *small_net)


for _n in [n for n in dir(P) if n.startswith('test_')]:
    globals()['test_hostsim__' + _n[5:]] = getattr(P, _n)


@pytest.mark.parametrize('pose,samples', [(0, 64), (2, 64), (0, 66)])
def test_hostsim_cull_fast_path_survivor_set(small_net, pose, samples):
    """The frame path's cull (k_front_cull<true>: cell mask + the pre-test + compacted candidates, csrc/front_bodies.h) only runs when
    the call has at least 4 ray-samples per lattice cell — more than the reduced frame of this module: a 128 x 128 frame of its own,
    every ray-sample against the dense stage kernels, at 64 and at 66 samples per ray (a thread's four samples then sit on different
    rays at different offsets).  (Inverting the pre-test's mask read fails this test; it does not fail
    on the 96 x 96 frame.)"""
    old = (P.RES, P.S)
    P.RES, P.S = 128, samples
    try:
        f = P.make_frame(pose, *small_net)
        assert f['gb']['ray_o'].shape[1] * P.S >= 4 * f['gb']['pbw'][0].shape[:3].numel()
        P.test_cull_survivor_set_exact_whole_frame(f)
    finally:
        P.RES, P.S = old


@pytest.mark.skipif(not os.environ.get('HOSTSIM_FULL'), reason='30 s on the wave machine: HOSTSIM_FULL=1 (tools/hostsim_asan.sh sets it); the GPU suite runs the same cases')
def test_hostsim_edge_cases(small_net):
    """tests/test_gpu_fullsize.py:test_edge_cases on the reduced frame: empty and ragged ray lists, sample counts that are not a
    multiple of the wave size, nothing surviving the cull, a survivor capacity that is too small (reported, and — what matters under
    tools/hostsim_asan.sh — no kernel writes past its arrays), rays in any order."""
    from invr import scene
    cfg, net = small_net
    bnp, _ = scene.make_scene(64, 64, seed=0, cam_dist=1.8)                 # (a smaller frame than the kernels' tests: nine whole renders below)
    gb = scene.to_torch(bnp)
    ctx = net.prepare(gb)
    ro, rd, nr, fa = (gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))
    n = ro.shape[0]
    e = torch.empty(0, 3)
    z = net.render_rays(ctx, e, e, torch.empty(0), torch.empty(0), 64)
    assert z['rgb_map'].shape == (0, 3) and z['acc_map'].shape == (0,)
    idx = torch.arange(0, 333) * (n // 333)
    for S in (2, 63, 65, 130):
        o = net.render_rays(ctx, ro[idx], rd[idx], nr[idx], fa[idx], S, want_raw=True)
        assert o['rgb_map'].shape == (333, 3) and o['raw'].shape == (333 * S, 4) and bool(torch.isfinite(o['rgb_map']).all())
        assert int(o['stats'][6]) == 0
    o = net.render_rays(ctx, ro[:200] + 50.0, rd[:200], nr[:200], fa[:200], 64, want_raw=True)
    assert int(o['stats'][0]) == 0 and float(o['rgb_map'].abs().max()) == 0 and float(o['raw'].abs().max()) == 0
    full = net.render_rays(ctx, ro, rd, nr, fa, 64, want_raw=True)
    na = int(full['stats'][0])
    assert na > 4096 and int(full['stats'][6]) == 0
    rgb, raw = full['rgb_map'].clone(), full['raw'].clone()
    for cap in (1000, 4096, na - 1):                       # too small: reported, nothing written out of bounds
        o = net.render_rays(ctx, ro, rd, nr, fa, 64, max_active=cap)
        assert int(o['stats'][6]) == 1, cap
    o = net.render_rays(ctx, ro, rd, nr, fa, 64, max_active=na, want_raw=True)          # exactly enough
    assert int(o['stats'][6]) == 0 and torch.equal(o['rgb_map'], rgb) and torch.equal(o['raw'], raw)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    o = net.render_rays(ctx, ro[perm], rd[perm], nr[perm], fa[perm], 64)
    assert torch.equal(o['rgb_map'], rgb[perm])
    for sl in (slice(0, n // 3), slice(n // 3, n)):
        o = net.render_rays(ctx, ro[sl], rd[sl], nr[sl], fa[sl], 64)
        assert torch.equal(o['rgb_map'], rgb[sl])


def test_hostsim_survivor_order_does_not_change_the_frame(tmp_path):
    """tests/test_gpu_fullsize.py:test_survivor_order_does_not_change_the_frame on a 96 x 96 x 64 frame, both processes on the wave
    machine (the full 512 x 512 x 128 frame was compared this way once, by hand: bit-identical)."""
    import tests.test_gpu_fullsize as F
    more = (8, 256) if os.environ.get('HOSTSIM_FULL') else ()          # (16 and 1024 were run once by hand: equal)
    a, b = F.order_frames(tmp_path, 'cpu', 96, ','.join(str(x) for x in (64,) + more), extra=('hostsim',))
    F.check_order_frames(a, b, 64, 12000)
    F.check_more_sample_counts(a, b, more)
