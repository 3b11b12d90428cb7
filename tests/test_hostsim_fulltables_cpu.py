"""CPU: the production encoder path of the FULL-SIZE model on the wave machine.  The 2^20-row tables of inb_377 take the 24-bit
reduction of the hash (hash_mod24), the x-corner delta fold (GridDev.xdelta) and the exact-reciprocal quotients — none of which a
2^12-row test model reaches (its prime 4099 fails the 24-bit bound: it runs the 32-bit path).  Here the whole 285,993,711-parameter
model (1.09 GB of tables, N(0, 0.1^2)) lives in host memory and a 64 x 64 x 64 frame goes through the host build of the kernel
sources: k_part_encode_rs_xcd against the generic encoder on every pair and against the oracle's hash_embed, the strict render test,
the two-phase MLP test — the bodies of tests/test_gpu_production_kernels.py."""
import copy

import pytest
import torch

import tests.test_gpu_production_kernels as P
from tests.hostsim import harness

SIZES = dict(RES=64, S=64, MIN=dict(na=5000, listed=5000, oracle_subset=500, oracle_chunk=500, enc_take=3000, enc_total=8000, enc_inside=1500,
                                    strict_rays=32, occ=30))


@pytest.fixture(scope='module', autouse=True)
def hostsim():
    old = {k: getattr(P, k) for k in ('DEV', 'RES', 'S', 'MIN')}
    P.DEV = 'cpu'
    P.RES, P.S, P.MIN = SIZES['RES'], SIZES['S'], SIZES['MIN']
    try:
        with harness.activate() as counters:
            yield counters
            assert counters.anomalies == 0, counters.anomalies
    finally:
        for k, v in old.items():
            setattr(P, k, v)


@pytest.fixture(scope='module')
def fr(hostsim):
    from invr.config import make_cfg
    from invr.network import Network
    torch.manual_seed(3)
    cfg = make_cfg(N_samples=SIZES['S'])
    net = Network(cfg=copy.deepcopy(cfg)).eval()
    assert sum(p.numel() for p in net.parameters()) == 285993711
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('embedder.dense') or name.endswith('embedder.hash'):
                p.normal_(0.0, 0.1, generator=g)
    return P.make_frame(0, cfg, net)


for _n in ('test_row_sum_xcd_encoder_vs_oracle_real_tables', 'test_render_strict_1e4_on_well_conditioned_pixels',
           'test_two_phase_mlp_and_winner_lists_vs_whole_field'):
    globals()['test_hostsim__' + _n[5:]] = getattr(P, _n)
