"""CPU: the PRODUCTION kernels of the eval frame (k_knn_pairs, k_warp_pairs + k_deform_pairs_slice, k_part_encode_rs_xcd, k_part_occ_all /
k_winner_lists / k_part_rgb_all) pinned directly — the test bodies of tests/test_gpu_production_kernels.py on a reduced frame
(96 x 96 pixels x 48 samples, 2^12-row tables) through the host build of the kernel sources (tests/hostsim).  The frame is large
enough for several 4096-slot groups (segmented pair / winner lists, the colour kernel's segment cursor) and for every class of the
KNN's lattice cells; the whole-frame sizes stay with -m gpu."""
import copy

import pytest
import torch

import tests.test_gpu_production_kernels as P
from tests.hostsim import harness

SMALL = dict(RES=96, S=48,
             MIN=dict(na=12000, listed=12000, oracle_subset=1500, oracle_chunk=500, enc_take=4000, enc_total=9000, enc_inside=2000,
                      strict_rays=48, occ=40))
POSE_IDS = [1, 2]          # pose_scale 1.0 at threshold 0.05, pose_scale 1.2 at the inb_lan threshold 0.1


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
