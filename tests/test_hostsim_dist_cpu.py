"""CPU, world_size 2 over gloo, with the REAL kernels (host build, tests/hostsim) on every rank:
  * eval: each rank renders its tile-cyclic shard of the golden frame, one all-gather, every rank holds the full frame — equal to the
    single-process render bit for bit and to the reference golden within 1e-4 (tests/test_dist_gloo.py checks the exchange with a
    stand-in renderer; here the product's renderer runs under it);
  * training: tests/test_gpu_dist_train.py's two-rank data-parallel run (fused backward, gradient arena, row-scalar all-reduces
    started during the backward, FusedAdam) against one process with the averaged gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tests.test_gpu_dist_train as D
from tests.hostsim import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _golden_setup():
    from invr import params, scene
    from invr.config import make_cfg
    from invr.network import Network
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'inb377_small.npz'))
    meta = dict(zip(g['meta_keys'].tolist(), g['meta_vals'].tolist()))
    cfg = make_cfg(table_log2=int(meta['table_log2']), N_samples=int(meta['n_samples']))
    sd = params.init_state_dict(cfg, seed=int(meta['param_seed']))
    bnp, _ = scene.make_scene(int(meta['H']), int(meta['W']), seed=int(meta['scene_seed']))
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    return g, cfg, net.eval(), scene.to_torch(bnp)


def _eval_worker(rank, world, port, tile, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        with harness.activate() as cnt:
            from invr import dist as idist
            g, cfg, net, batch = _golden_setup()
            ctx = net.prepare(batch)

            def render(ro, rd, nr, fa):
                o = net.render_rays(ctx, ro, rd, nr, fa, cfg.N_samples, want_raw=False)
                return o['rgb_map'], o['acc_map']
            rgb, acc = idist.render_frame(render, batch, rank, world, tile=tile)
            assert cnt.anomalies == 0
            torch.save({'rgb': rgb, 'acc': acc}, out_path + '.%d' % rank)
    finally:
        dist.destroy_process_group()


def test_two_rank_eval_shards_with_the_real_kernels(tmp_path):
    out = str(tmp_path / 'ev')
    mp.spawn(_eval_worker, args=(2, _free_port(), 64, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    assert torch.equal(r0['rgb'], r1['rgb']) and torch.equal(r0['acc'], r1['acc'])          # every rank holds the same full frame
    with harness.activate():
        g, cfg, net, batch = _golden_setup()
        o = net.render_rays(batch, batch['ray_o'][0], batch['ray_d'][0], batch['near'][0], batch['far'][0], cfg.N_samples, want_raw=False)
        assert torch.equal(o['rgb_map'], r0['rgb']) and torch.equal(o['acc_map'], r0['acc'])   # = the single-process frame, bit for bit
    assert float(np.abs(r0['rgb'].numpy() - g['render_rgb_map'].reshape(-1, 3)).max()) < 1e-4
    assert float(np.abs(r0['acc'].numpy() - g['render_acc_map'].reshape(-1)).max()) < 1e-4


def _train_worker(rank, world, port, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    D.DEV, D.STEPS = 'cpu', 2
    with harness.activate():
        D._rank_body(rank, out_path)


@pytest.mark.skipif(not os.environ.get('HOSTSIM_FULL'), reason='40 s on the wave machine: HOSTSIM_FULL=1; tests/test_gpu_dist_train.py is the same run on the GPU')
def test_two_rank_dp_training_with_the_real_kernels(tmp_path):
    old, old_steps = D.DEV, D.STEPS
    D.DEV, D.STEPS = 'cpu', 2
    try:
        with harness.activate():
            D.test_two_rank_dp_equals_single_rank_averaged_gradients(tmp_path, worker=_train_worker)
    finally:
        D.DEV, D.STEPS = old, old_steps
