"""GPU: the training half at BASELINE configs[3] / configs[4].

  * configs[4] shape — 1024 rays x 128 samples, full-size inb_377 model (1.09 GB tables): forward values and EVERY parameter
    gradient of one iteration (fused node, gradient arena, row-scalar table gradients) against CPU torch autograd of the oracle.
  * configs[3] — inb_lan.yaml training (smpl_thresh 0.1, pair_loss_weight 1e-4, lr 1e-3, ExponentialLR, bounds adoption at
    iter_step == 1, the reference's epoch / iteration schedule) through driver.train + FusedAdam: the loss trajectory over
    >= 12 optimisation steps against the same loop run on the oracle with torch.optim.Adam.
  * data-parallel training (configs[4], two ranks): see tests/test_gpu_dist_train.py.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

def _tol_report(line):
    """INVR_TOL_REPORT=<file>: the measured headroom of a widened tolerance, one line per check (tools/gpu.sh tol: five runs -> the bounds in
    the comments next to the asserts)"""
    import os
    f = os.environ.get('INVR_TOL_REPORT')
    if f:
        with open(f, 'a') as fh:
            fh.write(line + '\n')


from invr import scene, params, driver                 # noqa: E402
from invr.config import make_cfg                        # noqa: E402
from invr.network import Network                        # noqa: E402
from invr.trainer import NetworkWrapper                 # noqa: E402
from invr.optim import FusedAdam                        # noqa: E402
from tests import oracle_train as OT                    # noqa: E402

DEV = 'cuda:0'


def patch_batch(side, seed=0, frame=3, cam_dist=1.8, res=512, centre=(256, 256), pose_scale=0.5):
    y0, x0 = centre[0] - side // 2, centre[1] - side // 2
    bnp, _ = scene.make_scene(res, res, seed=seed, frame=frame, cam_dist=cam_dist, pose_scale=pose_scale, crop=(y0, x0, side, side))
    return scene.to_torch(bnp)


def test_configs4_full_size_iteration_vs_oracle_autograd(full_net):
    cfg0, net0 = full_net
    cfg = copy.deepcopy(cfg0)                                     # N_samples 128, inb_377 defaults
    net = net0                                                    # shared 286 M parameter model: restored below
    old_cfg, was_training = net.cfg, net.training
    net.cfg = cfg
    net.train()
    try:
        bc = patch_batch(32)
        assert bc['ray_o'].shape[1] == 1024
        gb = {k: v.to(DEV) for k, v in bc.items()}
        n, S = 1024, cfg.N_samples
        g = torch.Generator().manual_seed(5)
        jitter = torch.rand(n, S, generator=g)
        noise = torch.rand(n * S * 5, 3, generator=g)
        opt = FusedAdam([{'params': [p]} for p in net.parameters() if p.requires_grad], 5e-4, eps=1e-15).attach(net)
        opt.zero_grad()
        wrap = NetworkWrapper(net)
        wrap.renderer._jitter = lambda shape, device: jitter.to(device)
        wrap.renderer._pair_noise_dense = lambda rows, device: noise.to(device)[:rows]
        tb = dict(gb)
        tb['iter_step'] = 2
        ret, loss, stats, _ = wrap(tb, split='train')
        loss.backward()
        arena = opt.arena
        # ---- oracle: same objective, CPU autograd
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        leaves = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and dict(net.named_parameters())[k].requires_grad}
        ref_loss, ref_stats = OT.train_loss(sd, cfg, bc, jitter, noise)
        ref_loss.backward()
        assert abs(float(loss) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss))), (float(loss), float(ref_loss))
        for k in ('img_loss', 'reg_dist', 'offset_loss'):
            assert abs(float(stats[k]) - float(ref_stats[k])) < 2e-5 * max(1.0, abs(float(ref_stats[k]))), k
        if 'pair_loss' in ref_stats:                  # rows with |tocc - 0.5| < 0.02 exist (random tables: maybe none; the golden
            assert abs(float(stats['pair_loss']) - float(ref_stats['pair_loss'])) < 2e-5          # scene of test_gpu_parity has 345)
        else:
            assert float(stats['pair_loss']) == 0.0
        # ---- gradients: small tensors dense, part tables as row scalars (every row of the 68 M compared)
        named = dict(net.named_parameters())
        table_ids = {id(t) for t in arena.tables}
        checked = 0
        for k, ref_leaf in leaves.items():
            p = named[k]
            rg = ref_leaf.grad
            if id(p) in table_ids:
                continue
            got = arena.grad_of(p).cpu()
            assert rg is not None, k                 # every part network runs, also on zero points: zero gradients, never None
            if float(rg.abs().max()) == 0.0:         # (a part without any flagged pair)
                assert float(got.abs().max()) == 0.0, k
                checked += 1
                continue
            scale = max(float(rg.abs().max()), 1e-6)
            # deformer tensors: the pair term differentiates a difference of nearly equal unit vectors (arbitrated in float64 by
            # test_pair_term_gradient_float64_arbitration); with the fixture seeded (round 5) the error no longer moves from run to run —
            # five runs, round 6 (gpurun_out/r6tol, tools/gpu.sh tol): worst tensor mlp.4.weight 2.11e-3 .. 2.14e-3 of its scale,
            # mlp.2.bias 7.8e-4 .. 1.1e-3, every other one <= 3.8e-4 — so 5e-3 (2.3 x the worst seen; round 5 had let it out to 3e-2)
            tol = 5e-3 if k.startswith('tpose_deformer') else 2e-4
            if k.startswith('tpose_deformer'):
                _tol_report('configs4 %s: max |grad - oracle| / scale = %.3e (scale %.3e)' % (k, float((got - rg).abs().max()) / scale, scale))
            assert float((got - rg).abs().max()) <= tol * scale + 2e-7, (k, float((got - rg).abs().max()), scale)
            checked += 1
        for i, pn in enumerate(net.tpose_human.part_networks):
            e = pn.embedder
            q = 'tpose_human.part_networks.%d.embedder.' % i
            if float(leaves[q + 'hash'].grad.abs().max()) == 0.0 and float(leaves[q + 'dense'].grad.abs().max()) == 0.0:
                assert float(e.row_grad().abs().max()) == 0.0
                continue
            rows = torch.cat([leaves[q + 'dense'].grad.reshape(-1, 16), leaves[q + 'hash'].grad.reshape(-1, 16)], 0)
            # the dense gradient IS a row scalar broadcast (the CPU sums each feature column separately: equal up to rounding)
            assert float((rows - rows[:, :1]).abs().max()) <= 1e-5 * float(rows.abs().max()) + 1e-12
            ref_rows = rows[:, 0].to(DEV)
            got = e.row_grad()
            scale = max(float(ref_rows.abs().max()), 1e-6)
            err = (got - ref_rows).abs()
            assert float(err.max()) <= 2e-4 * scale + 2e-7, (i, float(err.max()), scale)
            touched = ref_rows != 0
            # rows the reference leaves at exactly 0 (never touched, or touched with an exactly-zero corner weight): nothing above
            # rounding dust (a corner weight of ~1e-8 instead of 0 times a gradient)
            assert float(got[~touched].abs().max()) <= 1e-7 * scale, (i, float(got[~touched].abs().max()), scale)
            assert int((got != 0).sum()) <= int(touched.sum()) + 1000
            checked += 1
        assert checked >= 40
    finally:
        net.cfg = old_cfg
        net.train(was_training)
        if hasattr(net, '_grad_arena'):
            del net._grad_arena
        for p in net.parameters():
            p.grad = None


def test_pair_term_gradient_float64_arbitration(small_setup, golden):
    """The pair regulariser (inb_renderer.py:78-94 + crit.py:8-18) differentiates || v_nb/|v_nb| - v_self/|v_self| || for residuals
    5 mm apart: a ~1e-3 difference of unit vectors, whose direction amplifies fp32 rounding ~1e3 x in ANY fp32 implementation.  So
    the fused backward's deformer gradients are arbitrated by a float64 run of the oracle: they must be as close to it as the
    oracle's own float32 run is (x4), with the loss made of the pair term alone (weight 10, inb_377.yaml)."""
    cfg0, sd0, batch, _ = small_setup                       # the golden scene: 345 of its 2155 dense rows have |tocc - 0.5| < 0.02
    cfg = copy.deepcopy(cfg0)
    cfg.pair_loss_weight = 10.0
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64))
    bc = dict(batch)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        bc[k] = batch[k][:, tsel]
    n, S = bc['ray_o'].shape[1], cfg.N_samples
    g = torch.Generator().manual_seed(41)
    jitter, noise = torch.rand(n, S, generator=g), torch.rand(n * S * 5, 3, generator=g)
    net = Network(cfg=copy.deepcopy(cfg))
    net.load_state_dict(sd0, strict=True)
    net = net.to(DEV).train()
    wrap = NetworkWrapper(net)
    wrap.renderer._jitter = lambda shape, device: jitter.to(device)
    wrap.renderer._pair_noise_dense = lambda rows, device: noise.to(device)[:rows]
    tb = {k: v.to(DEV) for k, v in bc.items()}
    tb['iter_step'] = 2
    ret, loss, stats, _ = wrap(tb, split='train')
    (cfg.pair_loss_weight * ret['pair_loss']).backward()          # (the differentiable scalar of the fused node; scalar_stats hold detached values)
    mine = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters() if k.startswith('tpose_deformer') and p.grad is not None}
    grads = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        leaves = [k for k in sd if k.startswith('tpose_deformer') and sd[k].is_floating_point() and k in mine]
        for k in leaves:
            sd[k].requires_grad_()
        b = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in bc.items()}
        _, st = OT.train_loss(sd, cfg, b, jitter.to(dt), noise.to(dt))
        (cfg.pair_loss_weight * st['pair_loss']).backward()
        grads[dt] = {k: sd[k].grad.double() for k in leaves}
        if dt == torch.float64:
            assert float(st['pair_loss']) > 1e-3 and abs(float(st['pair_loss']) - float(stats['pair_loss'])) < 2e-5
    assert len(mine) >= 7
    for k in mine:
        exact = grads[torch.float64][k]
        scale = float(exact.abs().max())
        e_mine, e_ref = float((mine[k] - exact).abs().max()), float((grads[torch.float32][k] - exact).abs().max())
        assert e_mine <= 4 * e_ref + 2e-4 * scale + 1e-9, (k, e_mine, e_ref, scale)


def test_lan_config_training_loop_vs_oracle():
    """BASELINE configs[3] (configs/inb/inb_lan.yaml over inb_377.yaml) at reduced table size: smpl_thresh 0.1,
    pair_loss_weight 1e-4, lr 1e-3 eps 1e-15, ExponentialLR per epoch, iter_step == 1 bounds adoption at the start of each
    epoch.  2 epochs x 7 iterations over varying frames / patches; the HIP path (driver.train, FusedAdam with gradient arena)
    against the oracle loop with torch.optim.Adam on the CPU."""
    cfg = make_cfg(table_log2=12, N_samples=32, smpl_thresh=0.1, pair_loss_weight=1e-4)
    LR, GAMMA, DECAY_EPOCHS, EPOCHS, EP_ITER = 1e-3, 0.1, 2, 2, 7          # decay_epochs 1000 in the yaml; 2 here so that the step shows
    sd0 = params.init_state_dict(cfg, seed=17)
    frames = [dict(seed=4, frame=10 + 7 * k, centre=(250 + 9 * k, 262 - 11 * k), pose_scale=0.5 + 0.05 * k) for k in range(5)]
    batches = [patch_batch(20, **kw) for kw in frames]
    g = torch.Generator().manual_seed(23)
    n_it = EPOCHS * EP_ITER
    jit = [torch.rand(b['ray_o'].shape[1], cfg.N_samples, generator=g) for b in batches]
    noi = [torch.rand(b['ray_o'].shape[1] * cfg.N_samples * 5, 3, generator=g) for b in batches]
    pick = lambda epoch, index: (epoch * EP_ITER + index) % len(batches)

    # ---- HIP path
    net = Network(cfg=copy.deepcopy(cfg))
    net.load_state_dict(sd0, strict=True)
    net = net.to(DEV).train()
    wrap = NetworkWrapper(net)
    opt = driver.make_optimizer(net, lr=LR, eps=1e-15)
    assert isinstance(opt, FusedAdam) and opt.arena is not None
    sched = driver.ExponentialLR(opt, decay_epochs=DECAY_EPOCHS, gamma=GAMMA)
    gbs = [{k: v.to(DEV) for k, v in b.items()} for b in batches]
    cur = {}

    def batch_fn(epoch, index):
        k = pick(epoch, index)
        cur['k'] = k
        return dict(gbs[k])
    wrap.renderer._jitter = lambda shape, device: jit[cur['k']].to(device)
    wrap.renderer._pair_noise_dense = lambda rows, device: noi[cur['k']].to(device)[:rows]
    out = driver.train(wrap, opt, batch_fn, EPOCHS, EP_ITER, scheduler=sched, stages=[{'_start': 0, 'ratio': 1.0}])
    assert out['iterations'] == n_it and net.cfg.ratio == 1.0
    mine = np.array(out['losses'])

    # ---- oracle loop (CPU)
    sd = {k: v.clone() for k, v in sd0.items()}
    train_keys = [k for k, p in Network(cfg=copy.deepcopy(cfg)).named_parameters() if p.requires_grad]
    for k in train_keys:
        sd[k].requires_grad_()
    ref_opt = torch.optim.Adam([{'params': [sd[k]], 'lr': LR} for k in train_keys], LR, eps=1e-15)
    ref_sched = driver.ExponentialLR(ref_opt, decay_epochs=DECAY_EPOCHS, gamma=GAMMA)
    ref = []
    for epoch in range(EPOCHS):
        for index in range(EP_ITER):
            k = pick(epoch, index)
            if index + 1 == 1:
                OT.adopt_batch_bounds(sd, cfg, batches[k])
            loss, _ = OT.train_loss(sd, cfg, batches[k], jit[k], noi[k])
            ref_opt.zero_grad(set_to_none=True)
            loss.backward()
            ref_opt.step()
            ref.append(float(loss))
        ref_sched.step()
    ref = np.array(ref)
    assert abs(opt.param_groups[0]['lr'] - ref_opt.param_groups[0]['lr']) < 1e-12 and opt.param_groups[0]['lr'] < LR * 0.4
    assert np.isfinite(mine).all()
    rel = np.abs(mine - ref) / np.abs(ref)
    assert rel[0] < 2e-5, rel[0]                                       # identical parameters: fp32 agreement of the objective
    assert rel.max() < 5e-3, (rel, mine, ref)                         # 14 Adam steps later (eps 1e-15: sign-like updates of noise-level gradients)
    # the adopted bounds are part of the checkpoint
    b0 = net.state_dict()['tpose_human.part_networks.0.embedder.bounds'].cpu()
    assert torch.equal(b0, sd['tpose_human.part_networks.0.embedder.bounds'])


def test_random_bg_epsilon_training_steps_vs_oracle():
    """cfg.random_bg True: inb_renderer.py:72 passes the flag to volume_rendering as render_weights' epsilon (net_utils.py:12-15, 18:
    weights = alpha * cumprod(1 - alpha + 1)); the fused forward / backward composite with that epsilon.  On a small model: every
    parameter gradient of one forward + backward against CPU autograd of the oracle, then three optimiser steps' losses against the
    oracle + torch.optim.Adam."""
    # (with epsilon = 1 the weights grow like 2^S and cancel in the sums: looser than the 2e-4 of the epsilon-0 golden gradients)
    _mode_training_steps(dict(random_bg=True), dict(random_bg=False), 1e-3, 3e-3)


def test_aggr_mean_training_steps_vs_oracle():
    """cfg.aggr = 'mean' (inb_part_network_multiassign.py:236-239) through NetworkWrapper + the fused node (k_merge_bwd<MEAN>) +
    FusedAdam: the same check — gradients of one iteration against CPU autograd of the (reference-pinned) oracle, three optimiser
    steps' losses against the oracle + torch.optim.Adam."""
    _mode_training_steps(dict(aggr='mean'), dict(aggr=''), 2e-4, 3e-3)


@pytest.mark.parametrize('aggr', ['dist', 'mindist'])
def test_aggr_distance_merges_training_steps_vs_oracle(aggr):
    """cfg.aggr = 'dist' / 'mindist' (inb_part_network_multiassign.py:240-251; round 5) through NetworkWrapper + the fused node
    (k_knn_pdist, k_winner_lists<2 / 3>, k_merge_bwd<2 / 0>) + FusedAdam: gradients of one iteration against CPU autograd of the
    (reference-pinned) oracle, three optimiser steps' losses against the oracle + torch.optim.Adam.  (The far parts' extrapolated
    field values dominate these merges: the looser bounds of the random_bg case.)"""
    _mode_training_steps(dict(aggr=aggr), dict(aggr=''), 1e-3, 3e-3)


def _mode_training_steps(over, default, tol, tol_deformer):
    cfg = make_cfg(table_log2=12, N_samples=12, **over)
    sd0 = params.init_state_dict(cfg, seed=21)
    bc = patch_batch(16, seed=2, frame=9, centre=(250, 262))
    n, S, LR, STEPS = bc['ray_o'].shape[1], 12, 1e-3, 3
    g = torch.Generator().manual_seed(3)
    jit, noi = torch.rand(n, S, generator=g), torch.rand(n * S * 5, 3, generator=g)
    gb = {k: v.to(DEV) for k, v in bc.items()}

    def fresh():
        net = Network(cfg=copy.deepcopy(cfg))
        net.load_state_dict(sd0, strict=True)
        net = net.to(DEV).train()
        wrap = NetworkWrapper(net)
        wrap.renderer._jitter = lambda shape, device: jit.to(device)
        wrap.renderer._pair_noise_dense = lambda rows, device: noi.to(device)[:rows]
        return net, wrap
    # ---- gradients of one forward + backward (no arena: ordinary dense gradients)
    net, wrap = fresh()
    b = dict(gb)
    b['iter_step'] = 2
    _, loss, _, _ = wrap(b, split='train')
    loss.mean().backward()
    mine_g = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters() if p.grad is not None}
    keys = [k for k, p in net.named_parameters() if p.requires_grad]
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in keys:
        sd[k].requires_grad_()
    l_ref, _ = OT.train_loss(sd, cfg, bc, jit, noi)
    l_ref.backward()
    assert abs(float(loss) - float(l_ref)) < 2e-5 * max(1.0, abs(float(l_ref)))
    cfg0 = copy.deepcopy(cfg)                                          # the switch is in play: with its default the loss is a different number
    cfg0.update(default)
    with torch.no_grad():
        l0, _ = OT.train_loss({k: v.detach() for k, v in sd0.items()}, cfg0, bc, jit, noi)
    assert abs(float(l0) - float(l_ref)) > 1e-3 * abs(float(l_ref))
    checked = 0
    for k in keys:
        gr = sd[k].grad
        if gr is None or k not in mine_g:
            continue
        scale = float(gr.abs().max())
        if scale == 0.0:
            assert float(mine_g[k].abs().max()) == 0.0, k
            continue
        err = float((mine_g[k] - gr.double()).abs().max())
        assert err <= (tol_deformer if k.startswith('tpose_deformer') else tol) * scale + 1e-12, (k, err, scale)
        checked += 1
    assert checked >= 40
    # ---- three optimiser steps: the loss trajectory
    net, wrap = fresh()
    opt = driver.make_optimizer(net, lr=LR, eps=1e-15)
    mine = [float(driver.train_step(wrap, opt, dict(gb), k + 2)[0]) for k in range(STEPS)]
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in keys:
        sd[k].requires_grad_()
    ref_opt = torch.optim.Adam([{'params': [sd[k]], 'lr': LR} for k in keys], LR, eps=1e-15)
    ref = []
    for k in range(STEPS):
        loss, _ = OT.train_loss(sd, cfg, bc, jit, noi)
        ref_opt.zero_grad(set_to_none=True)
        loss.backward()
        ref_opt.step()
        ref.append(float(loss))
    for a_, b_ in zip(mine, ref):
        assert abs(a_ - b_) < 5e-3 * max(1e-3, abs(b_)), (mine, ref)


def test_reference_step_form_with_disabled_grad_scaler(small_setup):
    """The reference's optimisation step, literally (lib/train/trainers/trainer.py:116-149): forward under
    autocast(enabled=cfg.use_amp), `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()` with
    GradScaler(enabled=cfg.use_amp), use_amp False — on the fused training node + gradient arena + FusedAdam.  The disabled scaler
    must be a pure pass-through: after 3 steps the parameters equal those of the plain `loss.backward(); optimizer.step()`
    sequence from the same start (up to the run-to-run rounding of the atomically accumulated gradients, see below)."""
    cfg, sd, batch, _ = small_setup
    g = torch.Generator().manual_seed(31)
    n = 256
    sel = torch.randperm(batch['ray_o'].shape[1], generator=g)[:n].sort()[0]
    b = dict(batch)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        b[k] = batch[k][:, sel]
    jit = torch.rand(n, cfg.N_samples, generator=g)
    noi = torch.rand(n * cfg.N_samples * 5, 3, generator=g)
    finals = []
    for mode in ('scaler', 'plain'):
        net = Network(cfg=copy.deepcopy(cfg))
        net.load_state_dict(sd, strict=True)
        net = net.to(DEV).train()
        wrap = NetworkWrapper(net)
        wrap.renderer._jitter = lambda shape, device: jit.to(device)
        wrap.renderer._pair_noise_dense = lambda rows, device: noi.to(device)[:rows]
        opt = driver.make_optimizer(net, lr=5e-4, eps=1e-15)
        assert isinstance(opt, FusedAdam) and opt.arena is not None
        scaler = torch.amp.GradScaler('cuda', enabled=False)
        for it in range(3):
            gb = {k: v.to(DEV) for k, v in b.items()}
            gb['iter_step'] = it + 2
            if mode == 'scaler':
                with torch.amp.autocast('cuda', enabled=False):
                    output, loss, loss_stats, image_stats = wrap(gb, 0, split='train')
                loss = loss.mean()
                opt.zero_grad(set_to_none=True)
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            else:
                output, loss, loss_stats, image_stats = wrap(gb, 0, split='train')
                loss = loss.mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
        assert np.isfinite(float(loss))
        finals.append({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
        steps = {int(st['step']) for st in opt.state_dict()['state'].values()}
        assert steps == {3}, steps                            # every tensor took every step (no part is skipped)
    # (not bit for bit: the backward accumulates weight / table gradients with float atomics, whose order differs from run to
    # run, and Adam with eps = 1e-15 turns a rounding-level gradient difference into a different step of the few elements whose
    # gradient IS rounding noise — so: no element further apart than the steps could move it, all but a sliver identical to 1e-6)
    moved = 0
    for k in finals[0]:
        a, b = finals[0][k].double(), finals[1][k].double()
        if not finals[0][k].is_floating_point():
            assert torch.equal(finals[0][k], finals[1][k]), k
            continue
        d = (a - b).abs()
        assert float(d.max()) <= 2 * 3 * 5e-4 * 1.01, (k, float(d.max()))
        n_off = int((d > 1e-6 + 1e-5 * b.abs()).sum())           # (a 192-element tensor came in with 2 such elements: 1 % of a small tensor is < 2)
        assert n_off <= max(4, 0.01 * d.numel()), (k, n_off, d.numel())
        moved += int(not torch.equal(finals[0][k], sd[k]))
    assert moved >= 60


def test_configs3_real_shape_three_steps_vs_oracle_autograd(full_net):
    """BASELINE configs[3] at its REAL shape: inb_lan.yaml over inb_377.yaml — smpl_thresh 0.1, pair_loss_weight 1e-4, lr 1e-3,
    eps 1e-15 — one 64 x 64 patch x 64 samples per iteration on the full-size model (285,993,711 parameters, 1.09 GB of tables),
    three optimiser steps through driver.train_step (the reference's step form) + FusedAdam, against the same three steps of CPU
    torch autograd of the oracle + torch.optim.Adam: the loss of every step, and after the third step every small tensor and the
    table rows the steps touched."""
    cfg0, net = full_net
    cfg = copy.deepcopy(cfg0)
    cfg.update(N_samples=64, smpl_thresh=0.1, pair_loss_weight=1e-4)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    old_cfg, was_training = net.cfg, net.training
    net.cfg = cfg
    net.train()
    LR, STEPS = 1e-3, 3
    try:
        bc = patch_batch(64, frame=11, centre=(250, 262))
        n, S = bc['ray_o'].shape[1], 64
        assert n == 4096
        g = torch.Generator().manual_seed(41)
        jit = [torch.rand(n, S, generator=g) for _ in range(STEPS)]
        noi = [torch.rand(n * S * 5, 3, generator=g) for _ in range(STEPS)]
        wrap = NetworkWrapper(net)
        cur = {'k': 0}
        wrap.renderer._jitter = lambda shape, device: jit[cur['k']].to(device)
        wrap.renderer._pair_noise_dense = lambda rows, device: noi[cur['k']].to(device)[:rows]
        opt = driver.make_optimizer(net, lr=LR, eps=1e-15)
        gb = {k: v.to(DEV) for k, v in bc.items()}
        mine = []
        for k in range(STEPS):
            cur['k'] = k
            loss, _ = driver.train_step(wrap, opt, dict(gb), k + 2)
            mine.append(float(loss))
            if k == 0:                  # the deformer after ONE step, for the float64 arbitration below
                got1 = {kk: v.detach().cpu().double() for kk, v in net.state_dict().items() if v.numel() <= (1 << 20)}      # (every small tensor)
        torch.cuda.synchronize()
        got = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        # ---- the oracle's steps on the CPU: three in float32 (the reference's arithmetic), ONE in float64 (the arbiter of the first
        # step, below: float64 over 286 M parameters is slow on the host)
        train_keys = [k for k, p in net.named_parameters() if p.requires_grad]
        runs, first = {}, {}
        for dt in (torch.float32, torch.float64):
            sdt = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
            for k in train_keys:
                sdt[k].requires_grad_()
            bdt = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in bc.items()}
            ref_opt = torch.optim.Adam([{'params': [sdt[k]], 'lr': LR} for k in train_keys], LR, eps=1e-15)
            losses = []
            for k in range(STEPS if dt == torch.float32 else 1):
                loss, _ = OT.train_loss(sdt, cfg, bdt, jit[k].to(dt), noi[k].to(dt), chunk=1024)
                ref_opt.zero_grad(set_to_none=True)
                loss.backward()
                ref_opt.step()
                losses.append(float(loss))
                if k == 0:
                    first[dt] = {kk: sdt[kk].detach().double().clone() for kk in train_keys if sdt[kk].numel() <= (1 << 20)}
            runs[dt] = ({k: sdt[k].detach().double() for k in train_keys} if dt == torch.float32 else None, losses)
            del ref_opt, sdt
        sd, ref = runs[torch.float32][0], runs[torch.float32][1]
        print('configs[3] real shape: losses', mine, 'oracle', ref)
        assert abs(mine[0] - ref[0]) < 2e-5 * max(1.0, abs(ref[0]))            # identical parameters: fp32 agreement of the objective
        for a, b in zip(mine, ref):
            assert abs(a - b) < 5e-3 * max(1e-3, abs(b)), (mine, ref)
        # parameters after three steps.  Adam with eps 1e-15 moves every element with a gradient by ~lr per step whatever the
        # gradient's size (sign-like), so elements whose gradient is rounding noise may differ by a few lr; everything else agrees.
        checked = 0
        for k in train_keys:
            a, b, o = got[k].double(), sd[k].detach().double(), sd0[k].double()
            moved_ref = (b - o).abs()
            if a.numel() > (1 << 22):                                           # part tables: the rows the three steps touched
                rows = (moved_ref.reshape(-1, a.shape[-1]).sum(1) > 0).nonzero(as_tuple=True)[0]
                if rows.numel() == 0:
                    assert torch.equal(got[k], sd0[k]), k
                    continue
                a, b = a.reshape(-1, a.shape[-1])[rows], b.reshape(-1, b.shape[-1])[rows]
                untouched = (moved_ref.reshape(-1, moved_ref.shape[-1]).sum(1) == 0)
                assert float((got[k].double() - o).abs().reshape(-1, o.shape[-1])[untouched].max()) <= 3.5 * LR, k   # noise-level rows at most
            d = (a - b).abs()
            frac_close = float((d <= 1e-5 + 1e-3 * b.abs()).double().mean())
            assert float(d.max()) <= 2 * STEPS * LR * 1.01, (k, float(d.max()))  # never more than the steps can move an element
            # The deformer's gradients carry the pair term's fp32 conditioning (test_pair_term_gradient_float64_arbitration): many of
            # its elements have gradients at rounding level, where Adam with eps 1e-15 steps by +-lr on the SIGN of noise — two fp32
            # evaluations of the same objective then disagree on those elements.  A float64 run of the oracle arbitrates: this
            # build must agree with it as often as the oracle's own float32 run does (the round-3 form of this assertion was a flat
            # floor on the fp32-vs-fp32 agreement, lowered from 0.75 to 0.5 when a run came in at 0.67: not shown to be noise).
            if k in first[torch.float64]:                        # every small tensor (the part tables are compared on their touched rows above)
                x = first[torch.float64][k]                      # (after the FIRST step: one Adam step = -lr * sign-like(gradient))
                close64 = lambda u: float(((u - x).abs() <= 1e-5 + 1e-3 * x.abs()).double().mean())
                f_mine, f_ref = close64(got1[k]), close64(first[torch.float32][k])
                if k.startswith('tpose_deformer'):
                    print('  %-40s step 1, agreement with the float64 oracle: HIP %.3f, float32 oracle %.3f; 3 steps, HIP vs float32 oracle %.3f'
                          % (k, f_mine, f_ref, frac_close))
                assert f_mine >= f_ref - 0.05, (k, f_mine, f_ref, frac_close)
                # three steps later, fp32 against fp32: a sanity floor only (both runs take sign-like steps on rounding-level gradients;
                # measured 0.88-1.00 for the deformer, 0.95-1.00 for the part MLPs with the pre-activation form of softplus')
                assert frac_close >= 0.8, (k, frac_close)
            else:
                assert frac_close >= 0.97, (k, frac_close)
            checked += 1
        assert checked >= 60
    finally:
        net.cfg = old_cfg
        net.train(was_training)
        if hasattr(net, '_grad_arena'):
            del net._grad_arena
        with torch.no_grad():
            for k, p in net.state_dict().items():
                p.copy_(sd0[k])
        for p in net.parameters():
            p.grad = None


def test_reference_built_adam_is_adopted_by_the_fused_step():
    """train_net.py unchanged: torch.optim.Adam built the reference's way (lib/train/optimizer.py:15-31: one group per tensor) over a
    network behind invr's NetworkWrapper, a scheduler built ON that optimizer (make_lr_scheduler), the reference's step form.
    At its first step() the optimizer object is bound to the fused step (invr.optim.adopt_on_first_step): it stays the SAME
    torch.optim.Adam object — the scheduler's lr writes are honoured, its state_dict has torch's layout and loads into a fresh
    torch.optim.Adam — and the run follows the FusedAdam + arena run built by driver.make_optimizer."""
    cfg = make_cfg(table_log2=12, N_samples=32, smpl_thresh=0.1, pair_loss_weight=1e-4)
    LR, EPOCHS, EP_ITER = 1e-3, 2, 4
    sd0 = params.init_state_dict(cfg, seed=19)
    batches = [patch_batch(20, seed=4, frame=10 + 7 * k, centre=(250 + 9 * k, 262 - 11 * k)) for k in range(3)]
    gbs = [{k: v.to(DEV) for k, v in b.items()} for b in batches]
    g = torch.Generator().manual_seed(29)
    jit = [torch.rand(b['ray_o'].shape[1], cfg.N_samples, generator=g).to(DEV) for b in batches]
    noi = [torch.rand(b['ray_o'].shape[1] * cfg.N_samples * 5, 3, generator=g).to(DEV) for b in batches]
    runs = {}
    for mode in ('reference_adam', 'fused'):
        net = Network(cfg=copy.deepcopy(cfg))
        net.load_state_dict(sd0, strict=True)
        net = net.to(DEV).train()
        wrap = NetworkWrapper(net)
        opt = driver.make_optimizer(net, lr=LR, eps=1e-15, fused=(mode == 'fused'))
        assert (type(opt) is torch.optim.Adam) == (mode == 'reference_adam')
        sched = driver.ExponentialLR(opt, decay_epochs=1, gamma=0.5)
        cur = {}

        def batch_fn(epoch, index, cur=cur):
            cur['k'] = (epoch * EP_ITER + index) % len(gbs)
            return dict(gbs[cur['k']])
        wrap.renderer._jitter = lambda shape, device, cur=cur: jit[cur['k']]
        wrap.renderer._pair_noise_dense = lambda rows, device, cur=cur: noi[cur['k']][:rows]
        out = driver.train(wrap, opt, batch_fn, EPOCHS, EP_ITER, scheduler=sched)
        torch.cuda.synchronize()
        runs[mode] = (np.array(out['losses']), {k: v.detach().cpu() for k, v in net.state_dict().items()}, opt, net, sched)
    l_ref, sd_ref, opt, net, sched = runs['reference_adam']
    l_fus, sd_fus = runs['fused'][:2]
    assert type(opt) is torch.optim.Adam and getattr(opt, '_invr_inner', None) is not None          # same object, adopted
    assert getattr(net, '_grad_arena', None) is opt._invr_inner.arena is not None                 # row-scalar table gradients from step 2 on
    assert opt._invr_inner.param_groups is opt.param_groups and opt._invr_inner.state is opt.state
    assert abs(opt.param_groups[0]['lr'] - LR * 0.5 ** EPOCHS) < 1e-12                            # the scheduler drove the object it was built on
    assert np.all(np.isfinite(l_ref)) and np.abs(l_ref - l_fus).max() <= 2e-4 * max(1.0, np.abs(l_fus).max()), (l_ref, l_fus)
    frac = []
    for k in sd_ref:
        if sd_ref[k].is_floating_point():
            d = (sd_ref[k].double() - sd_fus[k].double()).abs()
            assert float(d.max()) <= 2 * EPOCHS * EP_ITER * LR * 1.01, k          # (unordered float atomics + eps 1e-15: a few elements move by +-lr)
            if d.numel() >= 100000:          # the tables: all but a sliver of their rows agree to rounding; the small MLP tensors see every
                frac.append(float((d <= 1e-6 + 1e-5 * sd_fus[k].double().abs()).double().mean()))          # pair's noise (the loss curve pins them)
    assert len(frac) >= 5 and min(frac) >= 0.8, frac          # (measured 0.93 .. 1.0 run to run: rows whose gradient is rounding noise take +-lr steps under eps
                                                             #  1e-15, and the order of the backward's float atomics decides their sign; the loss curve above pins the run)
    osd = opt.state_dict()                                                                        # torch's layout, current step counts
    assert len(osd['state']) == len(opt.param_groups) and all(float(s['step']) == EPOCHS * EP_ITER for s in osd['state'].values())
    fresh = torch.optim.Adam([{'params': [p]} for p in net.parameters() if p.requires_grad], LR, eps=1e-15)
    fresh.load_state_dict(osd)                                                                    # net_utils.load_model's resume path
    assert float(fresh.state_dict()['state'][0]['step']) == EPOCHS * EP_ITER
    opt.load_state_dict(osd)                                                                      # ... and into the adopted object itself
    assert opt._invr_inner.state is opt.state
