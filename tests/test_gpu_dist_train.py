"""GPU: data-parallel training (BASELINE configs[4]; reference: DDP around NetworkWrapper, trainer.py:21-26).

Two ranks (one process each, torch.distributed) train on two DIFFERENT patches with invr.dist_train: full replicas,
gradients averaged over the ranks (DDP's semantics), FusedAdam with the gradient arena.  After k steps both replicas must
hold the parameters of ONE process that, each step, accumulates the gradients of the two patches with weight 1/2 (= the
averaged gradient) and steps once.  The box has one GPU: both ranks use cuda:0 and the gloo backend (RCCL refuses two
ranks on one device); the reducer code path is the production one except for the collective's transport."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, SIDE, SAMPLES = 4, 20, 32
DEV = 'cuda:0'                  # (tests/test_hostsim_dist_cpu.py runs the same bodies on the CPU wave machine)


def _setup():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import invr  # noqa: F401
    from invr import scene, params, driver
    from invr.config import make_cfg
    from invr.network import Network
    from invr.trainer import NetworkWrapper
    cfg = make_cfg(table_log2=12, N_samples=SAMPLES)
    sd = params.init_state_dict(cfg, seed=31)
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).train()
    batches = []
    for r, centre in enumerate(((250, 262), (270, 248))):
        bnp, _ = scene.make_scene(512, 512, seed=2, frame=11 + 40 * r, cam_dist=1.8, crop=(centre[0] - SIDE // 2, centre[1] - SIDE // 2, SIDE, SIDE))
        batches.append({k: v.to(DEV) for k, v in scene.to_torch(bnp).items()})
    g = torch.Generator().manual_seed(3)
    jit = [torch.rand(b['ray_o'].shape[1], SAMPLES, generator=g).to(DEV) for b in batches]
    noi = [torch.rand(b['ray_o'].shape[1] * SAMPLES * 5, 3, generator=g).to(DEV) for b in batches]
    wrap = NetworkWrapper(net)
    opt = driver.make_optimizer(net, lr=1e-3, eps=1e-15)
    return net, wrap, opt, batches, jit, noi, driver


def _worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _rank_body(rank, out_path)


def _rank_body(rank, out_path):
    try:
        net, wrap, opt, batches, jit, noi, driver = _setup()
        from invr import dist_train
        dist_train.broadcast_parameters(net)
        red = dist_train.attach(opt)
        assert red.world == 2
        wrap.renderer._jitter = lambda shape, device: jit[rank]
        wrap.renderer._pair_noise_dense = lambda rows, device: noi[rank][:rows]
        losses = []
        for it in range(STEPS):
            loss, _ = driver.train_step(wrap, opt, dict(batches[rank]), it + 2)
            losses.append(float(loss))
        torch.save({'sd': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'losses': losses}, out_path + '.%d' % rank)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_dp_equals_single_rank_averaged_gradients(tmp_path, worker=None):
    out = str(tmp_path / 'dp')
    mp.spawn(worker or _worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), k                     # replicas stay identical (same averaged gradient)
    # single process: gradients of the two patches accumulated with weight 1/2 in the arena, one step
    net, wrap, opt, batches, jit, noi, driver = _setup()
    init = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ref_losses = [[], []]
    for it in range(STEPS):
        opt.zero_grad(set_to_none=True)
        for r in range(2):
            wrap.renderer._jitter = lambda shape, device, r=r: jit[r]
            wrap.renderer._pair_noise_dense = lambda rows, device, r=r: noi[r][:rows]
            b = dict(batches[r])
            b['iter_step'] = it + 2
            ret, loss, stats, _ = wrap(b, split='train')
            (0.5 * loss.mean()).backward()
            ref_losses[r].append(float(loss.detach()))
        opt.step()
    ref = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    assert np.allclose(r0['losses'], ref_losses[0], rtol=2e-4) and np.allclose(r1['losses'], ref_losses[1], rtol=2e-4), (r0['losses'], ref_losses)
    moved = 0
    for k in ref:
        if not ref[k].is_floating_point():
            continue
        d_ref = (ref[k] - init[k]).abs().max()
        if float(d_ref) == 0:
            assert torch.equal(r0['sd'][k], ref[k]), k
            continue
        moved += 1
        # Adam with eps 1e-15 takes lr-sized steps along the SIGN of noise-level gradients: entries whose averaged gradient is
        # pure rounding noise may step the other way.  Everything else must agree to fp32 accuracy.
        diff = (r0['sd'][k] - ref[k]).abs()
        frac_bad = float((diff > 1e-5 + 1e-3 * d_ref).float().mean())
        assert frac_bad < 2e-3, (k, frac_bad, float(diff.max()), float(d_ref))
    assert moved >= 25                      # (parts without a flagged pair in these two patches have exactly zero gradients)
