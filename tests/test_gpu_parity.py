"""GPU parity: the HIP path, called through the C-ABI (libinvr.so), against the oracle and the
golden vectors of the imported reference.  fp32 tolerance stated per test; integer / index
results (cull mask, pflag) must match exactly except at measure-zero threshold ties, which are
detected through the oracle's decision margin and excluded explicitly."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nvr_oracle as O          # noqa: E402  (checker only)
from tests.conditioning import pixel_noise, part_field_noise  # noqa: E402  (checker only)
from invr import _abi, params               # noqa: E402
from invr.network import Network            # noqa: E402
from invr.renderer import Renderer          # noqa: E402

DEV = 'cuda:0'


def cu(x):
    return (torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(DEV, copy=True)


@pytest.fixture(scope='module')
def gpu_setup(small_setup):
    cfg, sd, batch, extras = small_setup
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    gb = {k: v.to(DEV) for k, v in batch.items()}
    return cfg, sd, batch, gb, net


class encoder_mode:
    """Eval renders read the part grids through row-sum tables by default (cfg.eval_row_sums); the
    parity tests run both that path and the direct 64-byte-row path."""

    def __init__(self, cfg, row_sums):
        self.cfg, self.val = cfg, row_sums

    def __enter__(self):
        self.old = self.cfg.get('eval_row_sums', True)
        self.cfg['eval_row_sums'] = self.val

    def __exit__(self, *a):
        self.cfg['eval_row_sums'] = self.old


BOTH_ENCODERS = pytest.mark.parametrize('row_sums', [True, False], ids=['rowsum', 'fullrow'])


def maxerr(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else b
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) if a.size else 0.0


def test_state_dict_layout(gpu_setup):
    cfg, sd, _, _, net = gpu_setup
    mine = net.state_dict()
    assert list(mine.keys()) == list(sd.keys())
    for k in sd:
        assert mine[k].shape == sd[k].shape and mine[k].dtype == sd[k].dtype, k


def test_sample_volume(gpu_setup, golden):
    cfg, sd, batch, gb, net = gpu_setup
    L = _abi.lib()
    pts = cu(golden['pose_pts'][0])
    vol = gb['pbw'][0].contiguous()
    dims = (C.c_int32 * 3)(*vol.shape[:3])
    out = torch.empty(pts.shape[0], 1, device=DEV)
    _abi.check(L.invr_sample_volume(_abi.ptr(vol), dims, vol.shape[3], vol.shape[3] - 1, 1,
                                    _abi.ptr(gb['pbounds'][0].contiguous()), _abi.ptr(pts), pts.shape[0],
                                    _abi.ptr(out), _abi.stream_ptr()))
    assert maxerr(out[:, 0], golden['pnorm']) < 1e-6
    up = cu(golden['uv_pts'][0])
    tuv = gb['tuv'][0].contiguous()
    dims = (C.c_int32 * 3)(*tuv.shape[:3])
    out = torch.empty(up.shape[0], 2, device=DEV)
    _abi.check(L.invr_sample_volume(_abi.ptr(tuv), dims, 2, 0, 2, _abi.ptr(gb['tbounds'][0].contiguous()),
                                    _abi.ptr(up), up.shape[0], _abi.ptr(out), _abi.stream_ptr()))
    assert maxerr(out.t()[None], golden['uv_out']) < 1e-6


def encoder_tolerance(xn, res, base=3e-6):
    """Per-(point, level) tolerance.  Outside the box the reference EXTRAPOLATES (offset measured from
    the clipped corner, part_base_embedder.py:117-118): the trilinear weights grow like
    prod_axis(1 + 2*cells_outside) and cancel, so fp32 summation-order noise is amplified by that
    factor.  Inside the box the factor is 1 and the tolerance is the plain fp32 one."""
    oob = np.maximum(np.maximum(-xn, xn - 1.0), 0.0)                         # (n,3) normalised units
    cells = oob[:, None, :] * (np.asarray(res, np.float32)[None, :, None] - 1)
    return base * np.prod(1.0 + 2.0 * cells, axis=-1) * 4.0                  # (n,L)


def test_grid_encoder_variants(gpu_setup, golden):
    cfg, sd, batch, gb, net = gpu_setup
    for tag, pid in (('body', 0), ('head', 2)):
        emb = net.tpose_human.part_networks[pid].embedder
        y = emb(cu(golden['emb_%s_x' % tag])).cpu().numpy()
        ref = golden['emb_%s_y' % tag]
        assert np.abs(y[:, :3] - ref[:, :3]).max() < 1e-6
        tol = encoder_tolerance(ref[:, :3], emb.spec['res'])
        assert (np.abs(y[:, 3:] - ref[:, 3:]) <= tol).all(), tag
        inside = (tol <= 1.3e-5).all(1)
        assert inside.sum() > 100 and np.abs(y[inside] - ref[inside]).max() < 1.3e-5
    emb = net.tpose_deformer.embedder
    y = emb(cu(golden['emb_deform_x'])).cpu().numpy()
    ref = golden['emb_deform_y']
    tol = np.repeat(encoder_tolerance(ref[:, :3], emb.spec['res']), 2, axis=1)
    assert (np.abs(y[:, 3:] - ref[:, 3:]) <= tol).all()
    # start_hash == 0: one (L,T,F) table, sum over levels
    kw = dict(n_levels=6, n_features_per_level=4, log2_hashmap_size=8, base_resolution=8, b=1.38,
              sum=True, sum_over_features=False, separate_dense=True, use_batch_bounds=False)
    sp = params.grid_spec(bbox=[[-1, -1, -1], [1, 2, 1]], **kw)
    tab = (np.random.RandomState(11).standard_normal((sp['L'], sp['T'], sp['F'])) * 0.1).astype(np.float32)
    keep = []
    g = _abi.make_grid(sp, None, cu(tab), cu(sp['bbox']), keep)
    x = cu(golden['emb_allhash_x'])
    out = torch.empty(x.shape[0], sp['out_dim'], device=DEV)
    _abi.check(_abi.lib().invr_grid_encode_fwd(C.byref(g), _abi.ptr(x), x.shape[0], _abi.ptr(out), _abi.stream_ptr()))
    xn = (golden['emb_allhash_x'] - sp['bbox'][0]) / (sp['bbox'][1] - sp['bbox'][0])
    tol = encoder_tolerance(xn, sp['res']).sum(1, keepdims=True)              # features are summed over levels
    assert (np.abs(out.cpu().numpy()[:, 3:] - golden['emb_allhash_y'][:, 3:]) <= tol).all()


def test_knn_blend(gpu_setup, golden):
    cfg, sd, batch, gb, net = gpu_setup
    keep = []
    scene = _abi.make_scene(gb, cfg, keep)
    ap = cu(golden['pose_pts'][0][golden['active_idx']])
    n = ap.shape[0]
    bw = torch.empty(n, 5, 24, device=DEV)
    dist = torch.empty(n, 5, device=DEV)
    _abi.check(_abi.lib().invr_knn_blend(C.byref(scene), _abi.ptr(ap), n, _abi.ptr(bw), _abi.ptr(dist), _abi.stream_ptr()))
    ref = golden['knn_bw'][0]
    assert maxerr(bw, ref[..., :24]) < 2e-6
    assert maxerr(dist, ref[..., 24]) < 2e-6
    margin = np.abs(ref[..., 24] - cfg.smpl_thresh)
    flag = (dist.cpu().numpy() < cfg.smpl_thresh)
    ok = (flag == golden['pflag'][0]) | (margin < 1e-6)
    assert ok.all()


def test_warp_deform(gpu_setup, golden):
    cfg, sd, batch, gb, net = gpu_setup
    keep = []
    scene = _abi.make_scene(gb, cfg, keep)
    model = net.model_struct(keep)
    act = golden['active_idx']
    ap = cu(golden['pose_pts'][0][act])
    S = cfg.N_samples
    rd = batch['ray_d'][0][torch.from_numpy(golden['sel_rays'].astype(np.int64))]
    pd = O.world_dirs_to_pose(rd[:, None].expand(-1, S, -1).reshape(-1, 3), batch['R'][0])
    pd = pd[torch.from_numpy(act.astype(np.int64))]
    bw = cu(golden['knn_bw'][0][..., :24].copy())
    flag = cu(golden['pflag'][0].astype(np.uint8))
    n = ap.shape[0]
    tp = torch.empty(n, 5, 3, device=DEV)
    td = torch.empty(n, 5, 3, device=DEV)
    rs = torch.empty(n, 5, 3, device=DEV)
    pdd = cu(pd).contiguous()                                   # (held: a temporary would be freed before the call reads it)
    _abi.check(_abi.lib().invr_warp_deform(C.byref(scene), C.byref(model), _abi.ptr(ap), _abi.ptr(pdd),
                                           _abi.ptr(bw), _abi.ptr(flag, torch.uint8), n, _abi.ptr(tp), _abi.ptr(td),
                                           _abi.ptr(rs), _abi.stream_ptr()))
    assert maxerr(tp[None], golden['tpose']) < 1e-5
    assert maxerr(td[None], golden['tpose_dirs']) < 1e-5
    assert maxerr(rs[None], golden['resd']) < 2e-6


@BOTH_ENCODERS
def test_part_fields(gpu_setup, golden, row_sums):
    cfg, sd, batch, gb, net = gpu_setup
    L = _abi.lib()
    keep = []
    with encoder_mode(cfg, row_sums):
        model = net.model_struct(keep)
    assert bool(model.part[0].grid.row_sums) == row_sums
    li = gb['latent_index'].reshape(-1)[:1].to(torch.int64).contiguous()
    pflag = golden['pflag'][0]
    n_inside = n_checked_outside = 0
    worst_outside = 0.0
    model64 = O.Model({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg)
    for pid in range(5):
        f = pflag[:, pid]
        tp = cu(golden['tpose'][0][f, pid].copy())
        td = cu(golden['tpose_dirs'][0][f, pid].copy())
        n = tp.shape[0]
        raw = torch.empty(n, 4, device=DEV)
        nb = L.invr_part_field_workspace(n)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _abi.check(L.invr_part_field_fwd(C.byref(model), pid, _abi.ptr(li, torch.int64), _abi.ptr(tp), _abi.ptr(td), n,
                                         _abi.ptr(raw), C.c_void_p(ws.data_ptr()), nb, _abi.stream_ptr()))
        # pairs whose canonical point lies outside the part's box are EXTRAPOLATED by the encoder (see
        # encoder_tolerance): that includes every "far" pair, which the reference's epsilon-normalised KNN
        # weights collapse onto the canonical origin.  fp32 noise is amplified there in the reference too.
        b = sd['tpose_human.part_networks.%d.embedder.bounds' % pid].numpy()
        xn = (golden['tpose'][0][f, pid] - b[0]) / (b[1] - b[0])
        inside = ((xn >= 0) & (xn <= 1)).all(1)
        err = np.abs(raw.cpu().numpy() - golden['part%d_raw' % pid]).max(1)
        n_inside += int(inside.sum())
        assert inside.sum() == 0 or err[inside].max() < 2e-5, pid
        # outside the box: the allowance follows the CONDITIONING of each point (tests/conditioning.py: the move of the float64 field
        # value under fp32-ulp perturbations of its inputs, and the reference's own fp32 deviation from float64), not a flat 1e-3
        _, noise = part_field_noise(O, model64, pid, torch.from_numpy(golden['tpose'][0][f, pid].copy()),
                                    torch.from_numpy(golden['tpose_dirs'][0][f, pid].copy()), int(li[0]),
                                    ref32=torch.from_numpy(golden['part%d_raw' % pid]))
        bound = 2e-5 + 8.0 * noise.numpy()
        assert (err <= bound).all(), (pid, float((err - bound).max()), float(err.max()))
        n_checked_outside += int((~inside).sum())
        worst_outside = max(worst_outside, float(err[~inside].max()) if (~inside).any() else 0.0)
    assert n_inside >= 100
    print('part fields: %d points outside their box, worst error %.2e (condition-aware bound)' % (n_checked_outside, worst_outside))


def test_composite_random(gpu_setup):
    g = torch.Generator().manual_seed(4)
    for R, S in ((1, 1), (7, 5), (33, 64), (20, 128), (9, 200)):
        raw = torch.rand(R, S, 4, generator=g)
        raw[..., 3] = raw[..., 3] * (torch.rand(R, S, generator=g) < 0.4)
        w, rgb, acc = O.composite(raw[..., :3], raw[..., 3])
        rg = cu(raw).contiguous()
        wo = torch.empty(R, S, device=DEV)
        ro = torch.empty(R, 3, device=DEV)
        ao = torch.empty(R, device=DEV)
        _abi.check(_abi.lib().invr_composite_fwd(_abi.ptr(rg), R, S, _abi.ptr(wo), _abi.ptr(ro), _abi.ptr(ao), _abi.stream_ptr()))
        assert maxerr(wo, w) < 2e-6 and maxerr(ro, rgb) < 5e-6 and maxerr(ao, acc) < 5e-6


def test_composite_backward_saturated_alphas(gpu_setup):
    """invr_composite_bwd vs torch autograd of volume_rendering / render_weights (net_utils.py:12-44, epsilon 0) with
    alphas of EXACTLY 0.0 and 1.0 in the ray (1 - exp(-softplus(h)) rounds to 1 for h >~ 17): torch's cumprod backward is
    zero-aware and stays finite there, and so must the kernel (no division by 1 - alpha)."""
    from invr import autograd as AG
    g = torch.Generator().manual_seed(9)
    for R, S in ((5, 1), (40, 64), (33, 65), (17, 200), (6, 700)):
        raw = torch.rand(R, S, 4, generator=g)
        u = torch.rand(R, S, generator=g)
        a = raw[..., 3].clone()
        a[u < 0.25] = 0.0
        a[(u > 0.9)] = 1.0                                   # several exact ones per ray, also consecutive / first / last
        a[0] = 0.0
        if S > 1:
            a[1, 0] = 1.0; a[2, -1] = 1.0; a[3, S // 2] = 1.0; a[3, S // 2 - 1:S // 2 + 1] = 1.0
            a[4] = torch.rand(S, generator=g) * 1e-3 + (1 - 1e-3)         # all close to one: catastrophic for a quotient form
        raw[..., 3] = a
        g_rgb, g_acc, g_w = torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, S, generator=g)
        rc = raw.clone().double().requires_grad_()
        w, rgb, acc = O.composite(rc[..., :3], rc[..., 3])
        ((rgb * g_rgb.double()).sum() + (acc * g_acc.double()).sum() + (w * g_w.double()).sum()).backward()
        rg = cu(raw).requires_grad_()
        wo, ro, ao = AG.CompositeFn.apply(rg)
        ((ro * cu(g_rgb)).sum() + (ao * cu(g_acc)).sum() + (wo * cu(g_w)).sum()).backward()
        got = rg.grad.cpu()
        assert bool(torch.isfinite(got).all()), (R, S)
        ref = rc.grad.float()
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-6 * max(scale, 1.0), (R, S, float((got - ref).abs().max()), scale)
        # and against torch's own fp32 autograd: same tolerance class
        rf = raw.clone().requires_grad_()
        w, rgb, acc = O.composite(rf[..., :3], rf[..., 3])
        ((rgb * g_rgb).sum() + (acc * g_acc).sum() + (w * g_w).sum()).backward()
        assert bool(torch.isfinite(rf.grad).all())
        assert float((got - rf.grad).abs().max()) <= 4e-6 * max(scale, 1.0)


@BOTH_ENCODERS
def test_render_64x64x32_vs_reference_golden(gpu_setup, golden, row_sums):
    """BASELINE config 1 through Renderer.render (eval): <= 1e-4 per pixel vs the reference."""
    cfg, sd, batch, gb, net = gpu_setup
    r = Renderer(net)
    with encoder_mode(cfg, row_sums):
        ret = r.render(dict(gb))
    # the image maps are on the host when render() returns, raw / occ (N x 20 bytes) follow on first access (LazyHostRet) —
    assert set(ret.pending()) == {'raw', 'occ'} and not ret['rgb_map'].is_cuda and not ret['acc_map'].is_cuda and len(ret) == 4
    assert 'raw' in ret and set(ret.pending()) == {'raw', 'occ'}
    assert set(ret.keys()) == {'rgb_map', 'acc_map', 'raw', 'occ'} and ret.pending() == ()
    assert all(not v.is_cuda for v in ret.values())                       # reference moves eval outputs to CPU
    assert ret['raw'].shape == (1, gb['ray_o'].shape[1] * cfg.N_samples, 4)
    assert ret['occ'].shape == (1, gb['ray_o'].shape[1] * cfg.N_samples, 1)
    stats = r.last_stats.cpu().numpy()
    assert stats[6] == 0
    assert int(stats[0]) >= int(golden['render_n_active_samples'])   # survivors of the cull >= samples with occ != 0
    err = np.abs(ret['rgb_map'].numpy() - golden['render_rgb_map']).max(-1)[0]
    assert int((err > 1e-4).sum()) == 0, float(err.max())
    # PSNR against the frame's target colours within 0.1 dB of the reference's (BASELINE metric; here it follows from the pixel bound)
    from invr.driver import psnr_metric
    gt = batch['rgb'][0].numpy().astype(np.float64)
    d_psnr = psnr_metric(ret['rgb_map'][0].numpy().astype(np.float64), gt) - psnr_metric(golden['render_rgb_map'][0].astype(np.float64), gt)
    assert abs(d_psnr) < 0.1, d_psnr
    assert maxerr(ret['acc_map'], golden['render_acc_map']) < 1e-4
    raw = ret['raw'][0].numpy()
    nz = golden['render_raw_nz_idx']
    assert np.abs(raw[nz] - golden['render_raw_nz']).max() < 1e-4
    mask = np.ones(raw.shape[0], bool)
    mask[nz] = False
    assert np.abs(raw[mask]).max() == 0.0                                 # untouched samples are exact zeros


def tocc_bound(golden, sd, cfg, latent_index):
    """Per-row allowance for the train-mode occupancies `tocc` (Na x P rows, part = row % 5): 2e-6 on rows the reference left at zero
    (unflagged pairs), 2e-5 + 8 x the conditioning of the occupancy at the row's canonical point (tests/conditioning.py) elsewhere —
    far pairs sit far outside their part's box, where the encoder extrapolates (see test_part_fields)."""
    model64 = O.Model({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg)
    ref = torch.from_numpy(golden['train_tocc'][0, :, 0].copy())
    tpose = torch.from_numpy(golden['train_tpts'][0] + golden['train_resd'][0])
    bound = torch.full(ref.shape, 2e-6, dtype=torch.float64)
    for pid in range(5):
        rows = torch.arange(pid, ref.shape[0], 5)
        rows = rows[ref[rows] != 0]
        if rows.numel():
            ref4 = torch.zeros(rows.numel(), 4)
            ref4[:, 3] = ref[rows]
            _, noise = part_field_noise(O, model64, pid, tpose[rows], torch.zeros(rows.numel(), 3), latent_index, ref32=ref4, column=3)
            bound[rows] = 2e-5 + 8.0 * noise
    return bound


def test_train_mode_forward_vs_reference_golden(gpu_setup, golden):
    """Train-mode forward (fixed jitter / pair noise): rgb_map, dense resd/tocc layouts, oresd and the
    distortion regulariser against the reference, and NetworkWrapper's loss assembly."""
    from invr.trainer import NetworkWrapper
    cfg, sd, batch, gb, net = gpu_setup
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64)).to(DEV)
    tb = dict(gb)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        tb[k] = gb[k][:, tsel]
    net.train()
    try:
        wrap = NetworkWrapper(net)
        r = wrap.renderer
        r._jitter = lambda shape, device: cu(golden['train_jitter'][0])
        r._pair_noise = lambda like: cu(golden['train_pair_u'])
        with torch.no_grad():
            ret, loss, stats, _ = wrap(tb, split='train')
    finally:
        net.eval()
    assert maxerr(ret['rgb_map'], golden['train_rgb_map']) < 1e-4
    assert maxerr(ret['acc_map'], golden['train_acc_map']) < 1e-4
    assert ret['resd'].shape == golden['train_resd'].shape and ret['tocc'].shape == golden['train_tocc'].shape
    assert maxerr(ret['resd'], golden['train_resd']) < 5e-6
    # tpts = init_bigpose of ALL Na x P rows (inb_part_network_multiassign.py:96-120,162-166), unflagged pairs included
    assert ret['tpts'].shape == golden['train_tpts'].shape and maxerr(ret['tpts'], golden['train_tpts']) < 1e-5
    terr = (ret['tocc'].detach().cpu().double()[0, :, 0] - torch.from_numpy(golden['train_tocc'][0, :, 0]).double()).abs()
    assert bool((terr <= tocc_bound(golden, sd, cfg, int(gb['latent_index'].reshape(-1)[0]))).all()), float(terr.max())   # condition-aware, not a flat 1e-3
    assert ret['oresd'].shape == golden['train_oresd'].shape
    assert maxerr(ret['oresd'], golden['train_oresd']) < 5e-6
    assert maxerr(ret['reg_distortion_loss'], golden['train_reg_distortion_loss']) < 1e-5
    # golden loss = mse + 0.1*dist + 0.1*offset (make_golden.py); ours adds pair_loss_weight*pair
    pair = float(stats['pair_loss'])
    mine = float(loss) - cfg.pair_loss_weight * pair
    assert abs(mine - float(golden['train_loss'])) < 1e-5


def _golden_grads(golden):
    out = {}
    for k in golden:
        if k.startswith('grad::'):
            out[k[6:]] = ('full', golden[k])
        elif k.startswith('grad_rows::'):
            name = k[11:]
            out[name] = ('rows', golden[k], golden['grad_vals::' + name], bool(golden['grad_full_equal::' + name]))
    return out


def _train_batch(gb, golden):
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64)).to(DEV)
    tb = dict(gb)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        tb[k] = gb[k][:, tsel]
    return tb


def _dense_pair_noise(net, tb, golden, cfg):
    """The fused path draws one uniform triple per dense (survivor, part) row; the reference draws rand_like of the SELECTED
    rows (golden['train_pair_u'], in dense-row order).  A gradient-free pass finds the selected rows, the golden draws are
    scattered to them."""
    r = Renderer(net)
    r._jitter = lambda shape, device: cu(golden['train_jitter'][0])
    r._pair_noise = lambda like: cu(golden['train_pair_u'])
    with torch.no_grad():
        ret0 = r.render(dict(tb))
    reg = ((ret0['tocc'].reshape(-1) - 0.5).abs() < 0.02).nonzero(as_tuple=True)[0]
    assert reg.numel() == golden['train_pair_u'].shape[1]
    dense = torch.zeros(tb['ray_o'].shape[1] * cfg.N_samples * 5, 3, device=DEV)
    dense[reg] = cu(golden['train_pair_u'][0])
    return dense


def _check_golden_grads(params, golden):
    ref = _golden_grads(golden)
    checked = 0
    for name, entry in ref.items():
        g = params[name].grad
        assert g is not None, name
        g = g.detach().cpu().numpy()
        if entry[0] == 'full':
            scale = max(float(np.abs(entry[1]).max()), 1e-6)
            assert np.abs(g - entry[1]).max() <= 2e-4 * scale + 2e-7, (name, float(np.abs(g - entry[1]).max()), scale)
        else:
            _, rows, vals, equal = entry
            flat = g.reshape(-1, g.shape[-1])
            scale = max(float(np.abs(vals).max()), 1e-6)
            got = flat[rows][:, :1] if equal else flat[rows]
            assert np.abs(got - vals).max() <= 2e-4 * scale + 2e-7, (name, float(np.abs(got - vals).max()), scale)
            mask = np.ones(flat.shape[0], bool)
            mask[rows] = False
            assert not mask.any() or np.abs(flat[mask]).max() <= 1e-6 * scale + 1e-9, name   # untouched rows stay ~0
        checked += 1
    assert checked >= 60
    for name, p in params.items():                           # nothing else got a gradient the reference lacks
        if p.grad is not None and name not in ref:
            assert float(p.grad.abs().max()) == 0.0, name


@pytest.mark.parametrize('mode', ['fused', 'fused_arena', 'graph'])
def test_train_step_gradients_vs_reference_golden(gpu_setup, golden, mode):
    """loss.backward() through the training forward: every parameter gradient of the reference (autograd of the PyTorch
    path, 256 rays, fixed jitter) to fp32 tolerance — through the fused node (invr_train_fwd / invr_train_bwd) with
    autograd-delivered dense gradients, through the same node with the persistent gradient arena (row-scalar table
    gradients, expanded for the comparison), and through the op-by-op autograd graph (cfg.train_fused False)."""
    import copy
    from invr.optim import FusedAdam
    cfg0, sd, batch, gb, net0 = gpu_setup
    net = copy.deepcopy(net0)
    net.cfg = copy.deepcopy(cfg0)
    net.cfg.train_fused = mode != 'graph'
    cfg = net.cfg
    tb = _train_batch(gb, golden)
    net.train()
    if mode == 'fused_arena':
        opt = FusedAdam([{'params': [p]} for p in net.parameters() if p.requires_grad], 1e-3, eps=1e-15).attach(net)
        opt.zero_grad()
    r = Renderer(net)
    r._jitter = lambda shape, device: cu(golden['train_jitter'][0])
    r._pair_noise = lambda like: cu(golden['train_pair_u'])
    if mode != 'graph':
        dense = _dense_pair_noise(net, tb, golden, cfg)
        r._pair_noise_dense = lambda rows, device: dense[:rows]
    net.zero_grad(set_to_none=True)
    ret = r.render(tb)
    assert maxerr(ret['rgb_map'], golden['train_rgb_map']) < 1e-4
    assert maxerr(ret['acc_map'], golden['train_acc_map']) < 1e-4
    assert maxerr(ret['reg_distortion_loss'], golden['train_reg_distortion_loss']) < 1e-5
    if mode == 'graph':
        offset = torch.norm(ret['resd'], dim=2).mean()
    else:
        offset = ret['offset_loss']                                  # reduced on the device; resd itself is read back lazily
        assert abs(float(offset) - float(np.linalg.norm(golden['train_resd'], axis=2).mean())) < 1e-7
        from invr.trainer import reg_raw_crit
        assert abs(float(ret['pair_loss']) - float(reg_raw_crit(torch.from_numpy(golden['train_oresd'])))) < 2e-5
    loss = ((ret['rgb_map'] - tb['rgb']) ** 2).mean() + 0.1 * ret['reg_distortion_loss'].mean() + 0.1 * offset   # make_golden.py's loss
    assert abs(float(loss.detach()) - float(golden['train_loss'])) < 1e-5
    loss.backward()
    # the reference's dynamic-shape outputs (lazy in the fused modes)
    assert ret['resd'].shape == golden['train_resd'].shape and maxerr(ret['resd'], golden['train_resd']) < 5e-6
    assert ret['tocc'].shape == golden['train_tocc'].shape
    terr = (ret['tocc'].detach().cpu().double()[0, :, 0] - torch.from_numpy(golden['train_tocc'][0, :, 0]).double()).abs()
    assert bool((terr <= tocc_bound(golden, sd, cfg, int(gb['latent_index'].reshape(-1)[0]))).all()), float(terr.max())   # condition-aware, not a flat 1e-3
    assert ret['oresd'].shape == golden['train_oresd'].shape and maxerr(ret['oresd'], golden['train_oresd']) < 5e-6
    if mode == 'fused_arena':
        assert all(pn.embedder.hash.grad is None for pn in net.tpose_human.part_networks)     # tables: row scalars only
        opt.arena.expand_tables()
    _check_golden_grads(dict(net.named_parameters()), golden)


def _mode_net(net0, cfg0, **over):
    import copy
    net = copy.deepcopy(net0)
    net.cfg = copy.deepcopy(cfg0)
    net.cfg.update(over)
    return net


AGGR_TAGS = [('mean', 'mean'), ('dist', 'dist'), ('mind', 'mindist')]


@pytest.mark.parametrize('tag,aggr', AGGR_TAGS)
def test_aggr_mean_render_vs_reference_golden(gpu_setup, golden, golden_modes, tag, aggr):
    """cfg.aggr = 'mean' (inb_part_network_multiassign.py:236-239: raw = the mean over the five parts, zeros for unflagged parts),
    'dist' (:240-244: parts weighted by normalize(1 / (part_dist + 1e-5))) and 'mindist' (:245-251: the part of smallest part_dist)
    through Renderer.render against the IMPORTED reference (tests/golden/make_golden_modes.py), both encoder paths."""
    cfg0, sd, batch, gb, net0 = gpu_setup
    net = _mode_net(net0, cfg0, aggr=aggr).eval()
    # The distance-weighted merges let the FAR parts dominate every survivor (their eps-normalised part_dist is ~0): their values come
    # from encoders evaluated far outside their boxes, where the reference extrapolates with weights of 1e3..1e9 that cancel — its own
    # fp32 result then carries that noise.  A pixel's allowance beyond the plain 1e-4 is 4 x the reference's OWN deviation from a
    # float64 run of the oracle (0 for all but a handful of pixels: p99 5e-7, one pixel 5e-5 in 'dist').
    noise_rgb = np.zeros(golden_modes[tag + '_rgb_map'].shape[1])
    noise_raw = np.zeros(golden_modes[tag + '_raw_nz'].shape[0])
    if aggr != 'mean':
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
        with torch.no_grad():
            r64 = O.render(O.Model(sd64, net.cfg), b64)
        noise_rgb = np.abs(golden_modes[tag + '_rgb_map'] - r64['rgb_map'].numpy()).max(-1)[0]
        noise_raw = np.abs(golden_modes[tag + '_raw_nz'] - r64['raw'][0].numpy()[golden_modes[tag + '_raw_nz_idx']]).max(-1)
        assert np.percentile(noise_rgb, 99) < 5e-6                     # (the allowance is about a handful of pixels)
    tie_ray = np.zeros(noise_rgb.shape[0], bool)
    tie_raw = np.zeros(noise_raw.shape[0], bool)
    if aggr == 'mindist':
        # A DISCRETE decision on ulp-level margins: the arg-min runs over the eps-normalised distances of FAR parts (1e-11 .. 1e-20, formed
        # from exp() deep in its underflow range); where the two smallest part_dist of a survivor agree to 1e-4 relative, the transcendental
        # units of two machines may order them differently; and where both are below 2e-30 they were formed from DENORMAL weights
        # (exp(-d^2 / 0.01125) < 1.2e-38: a few significant bits — measured on the MI355X: 1 survivor of 3607, part_dist 1.05e-36 vs the
        # reference's 1.50e-36 against a rival at 1.054e-36).  Such survivors — found from the float64 oracle's part_dist — and the rays
        # they sit on are exempt from the value comparison; everything else is held to it.
        S_ = int(net.cfg.N_samples)
        pts, _ = O.sample_points(b64['ray_o'], b64['ray_d'], b64['near'], b64['far'], S_)
        vd = b64['ray_d'][:, :, None].expand(-1, -1, S_, -1).reshape(-1, 3)
        with torch.no_grad():
            f64 = O.field(O.Model(sd64, net.cfg), pts.reshape(-1, 3), vd, b64, want_train=True)
        d2 = torch.sort(f64['dist'], dim=1)[0][:, :2]
        rel = (d2[:, 1] - d2[:, 0]) / d2[:, 1].clamp(min=1e-300)
        amb = ((rel <= 1e-4) | ((d2[:, 1] < 2e-30) & (rel <= 0.75))).numpy()          # (denormal weights carry 1..3 significant bits)
        amb_samples = f64['active'].numpy()[amb]
        assert amb.sum() <= 0.08 * max(len(amb), 1), int(amb.sum())
        tie_ray[np.unique(amb_samples // S_)] = True
        tie_raw = np.isin(golden_modes[tag + '_raw_nz_idx'], amb_samples)
    for row_sums in (True, False):
        r = Renderer(net)
        with encoder_mode(net.cfg, row_sums):
            ret = r.render(dict(gb))
        assert r.last_stats.cpu().numpy()[6] == 0
        err = np.abs(ret['rgb_map'].numpy() - golden_modes[tag + '_rgb_map']).max(-1)[0]
        assert int(((err > 1e-4 + 4 * noise_rgb) & ~tie_ray).sum()) == 0, float(err[~tie_ray].max())
        acc_err = np.abs(ret['acc_map'].numpy() - golden_modes[tag + '_acc_map'])[0]
        assert float(acc_err[~tie_ray].max()) < 1e-4 + 4 * float(noise_rgb.max())
        raw = ret['raw'][0].numpy()
        nz = golden_modes[tag + '_raw_nz_idx']
        assert ((np.abs(raw[nz] - golden_modes[tag + '_raw_nz']).max(-1) <= 1e-4 + 4 * noise_raw) | tie_raw).all()
        mask = np.ones(raw.shape[0], bool)
        mask[nz] = False
        assert np.abs(raw[mask]).max() == 0.0
        assert maxerr(ret['occ'][0, :, 0], raw[:, 3]) == 0.0
    assert np.abs(golden_modes[tag + '_rgb_map'] - golden['render_rgb_map']).max() > 1e-3          # (the switch does change the image)


@pytest.mark.parametrize('tag,aggr', AGGR_TAGS)
@pytest.mark.parametrize('mode', ['fused', 'graph'])
def test_aggr_mean_train_step_gradients_vs_reference_golden(gpu_setup, golden, golden_modes, mode, tag, aggr):
    """cfg.aggr = 'mean' / 'dist' / 'mindist' in train mode: forward, loss and every parameter gradient of the reference's autograd (256
    rays, fixed jitter) through the fused node (k_merge_bwd: g / 5 resp. g x the part's distance weight to every flagged part; the
    part of smallest part_dist) and through the op-by-op graph."""
    cfg0, sd, batch, gb, net0 = gpu_setup
    net = _mode_net(net0, cfg0, aggr=aggr, train_fused=(mode != 'graph'))
    cfg = net.cfg
    tb = _train_batch(gb, golden)
    net.train()
    r = Renderer(net)
    r._jitter = lambda shape, device: cu(golden['train_jitter'][0])
    r._pair_noise = lambda like: cu(golden_modes[tag + '_train_pair_u'])
    if mode != 'graph':
        sub = {'train_jitter': golden['train_jitter'], 'train_pair_u': golden_modes[tag + '_train_pair_u']}
        dense = _dense_pair_noise(net, tb, sub, cfg)
        r._pair_noise_dense = lambda rows, device: dense[:rows]
    net.zero_grad(set_to_none=True)
    ret = r.render(tb)
    assert maxerr(ret['rgb_map'], golden_modes[tag + '_train_rgb_map']) < 1e-4
    offset = torch.norm(ret['resd'], dim=2).mean() if mode == 'graph' else ret['offset_loss']
    loss = ((ret['rgb_map'] - tb['rgb']) ** 2).mean() + 0.1 * ret['reg_distortion_loss'].mean() + 0.1 * offset   # make_golden_modes.py's loss
    assert abs(float(loss.detach()) - float(golden_modes[tag + '_train_loss'])) < 1e-5
    loss.backward()
    assert ret['tocc'].shape == golden_modes[tag + '_train_tocc'].shape
    np.testing.assert_array_equal(golden_modes[tag + '_train_tocc'], golden['train_tocc'])           # the occupancies do not depend on the merge
    params_, checked = dict(net.named_parameters()), 0
    for key in golden_modes:
        if key.startswith(tag + '_grad::'):
            name, want = key[len(tag + '_grad::'):], golden_modes[key]
            g = params_[name].grad.detach().cpu().numpy()
            scale = max(float(np.abs(want).max()), 1e-6)
            assert np.abs(g - want).max() <= 2e-4 * scale + 2e-7, (name, float(np.abs(g - want).max()), scale)
        elif key.startswith(tag + '_grad_rows::'):
            name = key[len(tag + '_grad_rows::'):]
            rows, vals = golden_modes[key], golden_modes[tag + '_grad_vals::' + name]
            g = params_[name].grad.detach().cpu().numpy()
            flat = g.reshape(-1, g.shape[-1])
            scale = max(float(np.abs(vals).max()), 1e-6)
            assert np.abs(flat[rows][:, :1] - vals).max() <= 2e-4 * scale + 2e-7, (name, float(np.abs(flat[rows][:, :1] - vals).max()), scale)
            mask = np.ones(flat.shape[0], bool)
            mask[rows] = False
            assert not mask.any() or np.abs(flat[mask]).max() <= 1e-6 * scale + 1e-9, name
        else:
            continue
        checked += 1
    assert checked >= 60, checked


def test_train_pair_and_offset_term_gradients_fused_vs_graph(gpu_setup, golden):
    """The regulariser terms the goldens' loss does not contain (pair regulariser: neighbour deformer evaluations and
    crit.reg_raw_crit) — gradients of the fused node against torch autograd of the reference formulas on the op-by-op graph."""
    import copy
    cfg0, sd, batch, gb, net0 = gpu_setup
    tb = _train_batch(gb, golden)
    grads = {}
    for mode in ('graph', 'fused'):
        net = copy.deepcopy(net0).train()
        net.cfg = copy.deepcopy(cfg0)
        net.cfg.train_fused = mode == 'fused'
        from invr.trainer import NetworkWrapper
        wrap = NetworkWrapper(net)
        r = wrap.renderer
        r._jitter = lambda shape, device: cu(golden['train_jitter'][0])
        r._pair_noise = lambda like: cu(golden['train_pair_u'])
        if mode == 'fused':
            dense = _dense_pair_noise(net, tb, golden, net.cfg)
            r._pair_noise_dense = lambda rows, device: dense[:rows]
        tbb = dict(tb)
        tbb['iter_step'] = 2
        ret, loss, stats, _ = wrap(tbb, split='train')
        loss.backward()
        grads[mode] = ({k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}, float(loss), float(stats['pair_loss']))
    (ga, la, pa), (gf, lf, pf) = grads['graph'], grads['fused']
    assert abs(la - lf) < 2e-5 and abs(pa - pf) < 2e-5 and pa > 1e-3
    assert set(ga) == set(gf)
    # the pair term is ||v_nb/|v_nb| - v_self/|v_self||| of residuals 5 mm apart: a ~1e-3 difference of O(1) unit vectors, whose
    # direction (the gradient) carries the fp32 rounding of the operands amplified by ~1e3 in BOTH implementations
    for k in ga:
        scale = max(float(ga[k].abs().max()), 1e-6)
        tol = 2e-3 if k.startswith('tpose_deformer') else 2e-4
        assert float((ga[k] - gf[k]).abs().max()) <= tol * scale + 2e-7, (k, float((ga[k] - gf[k]).abs().max()), scale)


def test_network_wrapper_optimisation_steps(gpu_setup, golden):
    """Trainer.train's inner step (trainer.py:108-149) on the drop-in wrapper: forward, loss.mean(),
    zero_grad(set_to_none), backward, Adam step (one param group per tensor, eps 1e-15: optimizer.py:15-31);
    the loss must go down on a fixed patch."""
    import copy
    from invr.trainer import NetworkWrapper
    cfg, sd, batch, gb, net0 = gpu_setup
    net = copy.deepcopy(net0).train()
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64)).to(DEV)
    tb = dict(gb)
    for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy'):
        tb[k] = gb[k][:, tsel]
    wrap = NetworkWrapper(net)
    wrap.renderer._jitter = lambda shape, device: cu(golden['train_jitter'][0])
    groups = [{'params': [p], 'lr': 5e-3, 'weight_decay': 0} for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam(groups, lr=5e-3, eps=1e-15)
    losses = []
    for it in range(6):
        tb['iter_step'] = it + 2                      # (iter_step == 1 would re-create the bounds, embedder :107-109)
        ret, loss, stats, _ = wrap(tb, split='train')
        loss = loss.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        assert set(['reg_dist', 'offset_loss', 'img_loss', 'psnr', 'loss']) <= set(stats.keys())   # pair_loss only when pairs qualify
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_network_wrapper_lpips_patch_branch(gpu_setup, golden):
    """inb_trainer.py:188-214 (use_lpips, the inb_377.yaml default): the rays are put back on the patch's pixels through
    mask_at_box, the perceptual module gets (1,3,H,W) images and its value REPLACES the MSE in the loss."""
    import copy
    from invr.trainer import NetworkWrapper
    from invr.losses import PerceptualLoss
    cfg, sd, batch, gb, net0 = gpu_setup
    net = copy.deepcopy(net0).train()
    net.cfg = copy.deepcopy(cfg)
    net.cfg.use_lpips = True
    torch.manual_seed(3)
    pl = PerceptualLoss(allow_random=True)                 # built on the CPU: the wrapper moves it to the network's device
    wrap = NetworkWrapper(net, perceptual_loss=pl)
    assert next(wrap.perceptual_loss.parameters()).device.type == torch.device(DEV).type
    seen = {}

    class Spy(torch.nn.Module):
        def forward(self, x, t):
            seen['x'], seen['t'] = x, t
            return pl(x, t)
    wrap.perceptual_loss = Spy()
    tb = dict(gb)                                    # the whole 64x64 frame as the "patch": mask_at_box has holes
    tb['iter_step'] = 2
    ret, loss, stats, _ = wrap(tb, split='train')
    H, W = int(gb['H']), int(gb['W'])
    assert seen['x'].shape == (1, 3, H, W) and seen['t'].shape == (1, 3, H, W)
    m = gb['mask_at_box'][0].reshape(H, W)
    ref = torch.zeros(H, W, 3, device=DEV)
    ref[m] = ret['rgb_map'][0].detach()
    assert torch.equal(seen['x'][0].permute(1, 2, 0).detach(), ref)
    refg = torch.zeros(H, W, 3, device=DEV)
    refg[m] = gb['rgb'][0]
    assert torch.equal(seen['t'][0].permute(1, 2, 0), refg)
    assert {'lpips_loss', 'img_loss', 'psnr', 'loss'} <= set(stats)
    other = cfg.reg_dist_weight * stats['reg_dist'] + cfg.resd_loss_weight * stats['offset_loss']
    if 'pair_loss' in stats:
        other = other + cfg.pair_loss_weight * stats['pair_loss']
    assert abs(float(loss) - float(other + stats['lpips_loss'])) < 1e-6        # no separate MSE term (:206-209)
    loss.backward()
    gsum = sum(float(p.grad.abs().sum()) for p in net.parameters() if p.grad is not None)
    assert np.isfinite(gsum) and gsum > 0


def test_network_forward_on_points(gpu_setup, golden):
    """Network.forward(wpts, viewdir, dists, batch) on arbitrary world points == the per-sample raw/occ of
    the rendered frame (and therefore the reference's, see the render test)."""
    cfg, sd, batch, gb, net = gpu_setup
    S = cfg.N_samples
    pts, z = O.sample_points(batch['ray_o'], batch['ray_d'], batch['near'], batch['far'], S)
    wpts = pts.reshape(-1, 3)
    vd = batch['ray_d'][:, :, None].expand(-1, -1, S, -1).reshape(-1, 3)
    ret = net(cu(wpts), cu(vd.contiguous()), None, gb)
    assert ret['raw'].shape == (1, wpts.shape[0], 4) and ret['occ'].shape == (1, wpts.shape[0], 1)
    raw = ret['raw'][0].cpu().numpy()
    nz = golden['render_raw_nz_idx']
    assert np.abs(raw[nz] - golden['render_raw_nz']).max() < 1e-4
    mask = np.ones(raw.shape[0], bool)
    mask[nz] = False
    assert np.abs(raw[mask]).max() == 0.0
    assert maxerr(ret['occ'][0, :, 0], raw[:, 3]) == 0.0


@pytest.mark.parametrize('over', [dict(smpl_thresh=0.1, N_samples=24),            # inb_lan.yaml threshold
                                  dict(smpl_thresh=1e9, N_samples=8),              # dense stress: every sample active
                                  dict(smpl_thresh=0.02, N_samples=40),
                                  dict(random_bg=True, N_samples=12),              # inb_renderer.py:72: the flag becomes render_weights' epsilon (= 1)
                                  dict(aggr='mean', N_samples=16),                 # inb_part_network_multiassign.py:236-239
                                  dict(aggr='mean', smpl_thresh=0.1, N_samples=12)])
def test_render_config_variants_vs_oracle(small_setup, over):
    """Hot-path flags other than the inb_377 defaults, against the (reference-pinned) oracle."""
    from invr.config import make_cfg
    _, _, batch, _ = small_setup
    cfg = make_cfg(table_log2=12, **over)
    sd = params.init_state_dict(cfg, seed=11)
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    sel = torch.arange(0, batch['ray_o'].shape[1], 7)
    b = dict(batch)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = batch[k][:, sel]
    gb = {k: v.to(DEV) for k, v in b.items()}
    r = Renderer(net)
    ret = r.render(dict(gb))
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b)
    stats = r.last_stats.cpu().numpy()
    assert stats[6] == 0
    nz_ref = (ref['raw'][0, :, 3] != 0).numpy()
    nz = (ret['raw'][0, :, 3] != 0).numpy()
    assert (nz == nz_ref).all()
    # values: 1e-4, except pixels whose reference arithmetic is itself ill-conditioned (extrapolated far pairs);
    # arbitrated by a float64 run of the oracle, as in test_gpu_fullsize.py
    with torch.no_grad():
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        ref64 = O.render(O.Model(sd64, cfg), b64)
    exact = ref64['rgb_map'][0]
    err_gpu = (ret['rgb_map'][0].double() - exact).abs().max(1)[0]
    err_ref = pixel_noise(O, O.Model(sd64, cfg), b64, exact, int(cfg.N_samples), chunk=4096, ref32=ref['rgb_map'][0], trials=4)      # tests/conditioning.py
    scale = max(1.0, float(exact.abs().max()))            # (epsilon = 1 lets the weights grow like 2^S: the bars scale with the image)
    assert bool((err_gpu <= 1e-4 * scale + 4 * err_ref).all()), (float(err_gpu.max()), float(err_ref.max()), scale)
    assert float(err_gpu.median()) < 5e-6 * scale
    if over.get('random_bg'):
        assert scale > 4.0                                # the epsilon did change the compositing


def test_knn_fallback_when_vertex_sets_exceed_lds_index(small_setup):
    """More posed vertices than the LDS-resident KNN index holds (8192 slots over the five parts; SMPL has 6890, SMPL-X 10475):
    the render falls back to the brute-force pair kernel (k_knn_pairs_bf) on the device, without a host decision.  The per-part
    reference sets are doubled with slightly displaced copies (13780 vertices, largest part 7378) and the render is checked against
    the oracle like every other scene."""
    from invr.config import make_cfg
    _, _, batch, _ = small_setup
    cfg = make_cfg(table_log2=12, N_samples=24)
    sd = params.init_state_dict(cfg, seed=13)
    b = dict(batch)
    pp, pw, l2 = batch['part_pts'][0], batch['part_pbw'][0], batch['lengths2'][0]
    M = int(l2.max())
    g = torch.Generator().manual_seed(2)
    npp, npw = torch.zeros(5, 2 * M, 3), torch.zeros(5, 2 * M, 24)
    for p in range(5):
        n = int(l2[p])
        npp[p, :n], npw[p, :n] = pp[p, :n], pw[p, :n]
        npp[p, n:2 * n] = pp[p, :n] + (torch.rand(n, 3, generator=g) - 0.5) * 2e-3
        npw[p, n:2 * n] = pw[p, :n]
    b['part_pts'], b['part_pbw'], b['lengths2'] = npp[None], npw[None], (2 * l2)[None]
    assert int(b['lengths2'].sum()) > 8192 and 2 * M <= 8192
    sel = torch.arange(0, batch['ray_o'].shape[1], 9)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = batch[k][:, sel]
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    r = Renderer(net)
    ret = r.render({k: v.to(DEV) for k, v in b.items()})
    assert int(r.last_stats[6]) == 0
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        exact = O.render(O.Model(sd64, cfg), b64)['rgb_map'][0]
    assert ((ret['raw'][0, :, 3] != 0) == (ref['raw'][0, :, 3] != 0)).all()
    assert int((ref['raw'][0, :, 3] != 0).sum()) > 100
    err_gpu = (ret['rgb_map'][0].double() - exact).abs().max(1)[0]
    err_ref = pixel_noise(O, O.Model(sd64, cfg), b64, exact, int(cfg.N_samples), chunk=4096, ref32=ref['rgb_map'][0], trials=4)      # tests/conditioning.py
    assert bool((err_gpu <= 1e-4 + 4 * err_ref).all()), (float(err_gpu.max()), float(err_ref.max()))
    assert float(err_gpu.median()) < 5e-6


def test_tpose_viewdir_false_is_rejected(gpu_setup):
    """cfg.tpose_viewdir=False cannot run in the reference either (TPoseHuman.forward indexes the (Na,3)
    view-dir tensor per part, inb_part_network_multiassign.py:216); the library reports it instead of
    silently rendering something else."""
    import copy
    cfg, sd, batch, gb, net = gpu_setup
    net2 = copy.deepcopy(net)
    net2.cfg = copy.deepcopy(cfg)
    net2.cfg.tpose_viewdir = False
    with pytest.raises(RuntimeError, match='tpose_viewdir'):
        net2.render_rays(gb, gb['ray_o'][0][:64], gb['ray_d'][0][:64], gb['near'][0][:64], gb['far'][0][:64], 16)


def test_run_evaluate_psnr_within_0p1_db(gpu_setup, golden):
    """The eval loop of run.py:61-90 on the drop-in classes: PSNR (Evaluator.psnr_metric, whole HxW image)
    of our render vs the PSNR of the reference's render of the same frame: BASELINE asks for 0.1 dB."""
    from invr import driver
    cfg, sd, batch, gb, net = gpu_setup
    res = driver.run_evaluate(net, [batch], device=DEV)
    gt = driver.assemble_image(batch['rgb'][0].numpy(), batch)
    ref = driver.assemble_image(golden['render_rgb_map'][0], batch)
    ref_psnr = driver.psnr_metric(ref.reshape(-1, 3), gt.reshape(-1, 3))
    assert abs(res['psnr'][0] - ref_psnr) < 1e-3, (res['psnr'][0], ref_psnr)


def test_generate_rays_vs_reference_golden():
    """invr_generate_rays against the reference's get_rays_within_bounds (tests/golden/rays_small.npz):
    integer/byte results (mask) exact; float32 rays/near/far bit-exact."""
    import os
    from invr import rays
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rays_small.npz'))
    for tag in ('a', 'b'):
        H, W = (int(v) for v in g[tag + '_HW'])
        ro, rd, near, far, mask = rays.rays_within_bounds(H, W, g[tag + '_K'], g[tag + '_R'], g[tag + '_T'], g[tag + '_bounds'], DEV)
        assert np.array_equal(mask.cpu().numpy(), g[tag + '_mask'])
        assert np.array_equal(ro.cpu().numpy(), g[tag + '_ray_o'])
        assert np.array_equal(rd.cpu().numpy(), g[tag + '_ray_d'])
        assert np.array_equal(near.cpu().numpy(), g[tag + '_near']) and np.array_equal(far.cpu().numpy(), g[tag + '_far'])


def test_scene_prep_on_device(small_setup):
    """Row f4: get_rigid_transformation vs the reference goldens (tests/golden/rigid_small.npz) and the
    per-part packing vs the dataset logic restated in invr.scene (bit-exact: it only moves data)."""
    import os
    from invr import prep
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rigid_small.npz'))
    for tag in ('p', 'q', 'z'):
        A = prep.rigid_transformation(cu(g[tag + '_poses']), cu(g[tag + '_joints']), cu(g['parents']))
        ref = g[tag + '_A']
        err = np.abs(A.cpu().numpy() - ref)
        assert err.max() <= 2.4e-7 and (err > 0).sum() <= 4          # float32 cast of float64 results: bit-equal up to rare 1-ulp flips
    cfg, sd, batch, extras = small_setup
    pp, pb, l2, bd = prep.pack_parts(cu(batch['ppts'][0]), cu(extras['weights']), cu(extras['parts']), cu(extras['tpose']), 0.2)
    assert torch.equal(l2.cpu(), batch['lengths2'][0])
    assert torch.equal(pp.cpu(), batch['part_pts'][0]) and torch.equal(pb.cpu(), batch['part_pbw'][0])
    assert torch.equal(bd.cpu(), batch['bounds'][0])
    # and against the reference's own inline code (tpose_dataset.py:569-591, executed by tests/golden/make_golden_parts.py)
    from tests.test_oracle_golden import parts_inputs, check_parts_against_golden
    gp = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'parts_small.npz'))
    for tag, seed in (('a', 0), ('b', 7)):
        ppts, weights, parts, tpose = parts_inputs(seed)
        pp, pb, l2, bd = prep.pack_parts(cu(ppts), cu(weights), cu(parts), cu(tpose), float(gp[tag + '_overlap']))
        check_parts_against_golden(tag, pp.cpu().numpy(), pb.cpu().numpy(), l2.cpu().numpy(), bd.cpu().numpy(), gp)


def test_pair_deformer_matches_point_deformer(gpu_setup):
    """The pair-list deformer of the render pipeline (MFMA MLP, per-frame t-slices of the grid held in LDS)
    against the thread-per-point deformer behind Network.resd (3-D grid lookups, VALU MLP), which
    test_warp_deform pins to the reference goldens."""
    cfg, sd, batch, gb, net = gpu_setup
    ro, rd, nr, fr = (gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))
    out = net.render_rays(gb, ro, rd, nr, fr, cfg.N_samples)
    torch.cuda.synchronize()
    v = _abi.ws_views(*out['_ws'])
    stats = out['stats'].cpu().numpy()
    n_checked = 0
    for p in range(5):
        c = int(stats[1 + p])
        r_pairs = v['l_r'][p][:, :c].t().contiguous()
        xb = (v['l_x'][p][:, :c].t() - r_pairs).contiguous()                # init_bigpose up to 1 ulp
        r_pts = net.resd(xb[None], gb)[0]
        assert maxerr(r_pairs, r_pts) < 1e-6, p
        assert float(r_pairs.abs().max()) <= 0.05 and float(r_pairs.abs().max()) > 1e-4
        n_checked += c
    assert n_checked > 1000


def test_fused_adam_matches_torch_adam():
    """Row f1: invr_adam_step (one launch for all tensors) vs torch.optim.Adam with the reference's construction
    (one parameter group per tensor, eps 1e-15), including tensors that get no gradient in some steps, per-group
    learning-rate changes (the reference's schedulers write group['lr']) and state_dict interchange."""
    from invr.optim import FusedAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(3,), (64, 70), (17,), (5, 4099, 16), (100000,), (1,), (33, 7)]
    ref_p = [torch.randn(s, generator=g).to(DEV).requires_grad_() for s in shapes]
    my_p = [p.detach().clone().requires_grad_() for p in ref_p]
    mk = lambda ps: [{'params': [p], 'lr': 5e-4, 'weight_decay': 0.0} for p in ps]
    ref = torch.optim.Adam(mk(ref_p), 5e-4, eps=1e-15)
    mine = FusedAdam(mk(my_p), 5e-4, eps=1e-15)
    for it in range(6):
        for k, (a, b) in enumerate(zip(ref_p, my_p)):
            if (it + k) % 4 == 3:                       # no gradient for this tensor in this step
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** ((k % 3) - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 3:
            for opt in (ref, mine):
                for grp in opt.param_groups:
                    grp['lr'] *= 0.5
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        assert maxerr(a, b) <= 2e-6 * float(a.detach().abs().max()) + 1e-9
    sa, sb = ref.state_dict(), mine.state_dict()
    assert sa['param_groups'][0].keys() >= {'lr', 'betas', 'eps', 'weight_decay'} and len(sa['state']) == len(sb['state'])
    for k in sa['state']:
        assert float(sa['state'][k]['step']) == float(sb['state'][k]['step'])
        assert maxerr(sa['state'][k]['exp_avg'], sb['state'][k]['exp_avg']) <= 1e-6 * float(sa['state'][k]['exp_avg'].abs().max()) + 1e-12
        # (1 - beta2 is formed in double: 1 - 0.999f is 1.3e-5 off 0.001 — what this line caught)
        assert maxerr(sa['state'][k]['exp_avg_sq'], sb['state'][k]['exp_avg_sq']) <= 1e-6 * float(sa['state'][k]['exp_avg_sq'].abs().max()) + 1e-12
    ref2 = torch.optim.Adam(mk([p.detach().clone().requires_grad_() for p in my_p]), 5e-4, eps=1e-15)
    ref2.load_state_dict(mine.state_dict())             # checkpoints interchange


def test_fuse_continues_a_reference_built_adam():
    """invr.optim.fuse: an Adam built the reference's way (lib/train/optimizer.py:13-31) that has already taken steps is replaced by
    a FusedAdam over the same groups and state; both continue for three steps from the same gradients and must agree."""
    from invr.optim import FusedAdam, fuse
    g = torch.Generator().manual_seed(12)
    shapes = [(3,), (64, 70), (5, 4099, 16), (33, 7)]
    ref_p = [torch.randn(s, generator=g).to(DEV).requires_grad_() for s in shapes]
    my_p = [p.detach().clone().requires_grad_() for p in ref_p]
    mk = lambda ps: [{'params': [p], 'lr': 5e-4 * (1 + k), 'weight_decay': 0.0} for k, p in enumerate(ps)]
    ref, mine = torch.optim.Adam(mk(ref_p), 5e-4, eps=1e-15), torch.optim.Adam(mk(my_p), 5e-4, eps=1e-15)
    sched = torch.optim.lr_scheduler.ExponentialLR(mine, 0.9)             # (adds initial_lr to the groups)

    def grads():
        for a, b in zip(ref_p, my_p):
            gr = torch.randn(a.shape, generator=g).to(DEV) * 0.1
            a.grad, b.grad = gr.clone(), gr.clone()
    for _ in range(2):
        grads(); ref.step(); mine.step()
    fused = fuse(mine)
    assert isinstance(fused, FusedAdam) and fuse(fused) is fused
    assert [grp['lr'] for grp in fused.param_groups] == [grp['lr'] for grp in mine.param_groups]
    assert all('initial_lr' in grp for grp in fused.param_groups) and sched is not None
    assert fuse(torch.optim.SGD(mk(my_p), 0.1)).__class__ is torch.optim.SGD                          # not an Adam: unchanged
    assert fuse(torch.optim.Adam(mk(my_p), 5e-4, amsgrad=True)).__class__ is torch.optim.Adam         # unsupported variant: unchanged
    for _ in range(3):
        grads(); ref.step(); fused.step()
    for a, b in zip(ref_p, my_p):
        assert maxerr(a, b) <= 2e-6 * float(a.detach().abs().max()) + 1e-9
    sa, sb = ref.state_dict(), fused.state_dict()
    for k in sa['state']:
        assert float(sa['state'][k]['step']) == float(sb['state'][k]['step']) == 5.0
        assert maxerr(sa['state'][k]['exp_avg_sq'], sb['state'][k]['exp_avg_sq']) <= 1e-6 * float(sa['state'][k]['exp_avg_sq'].abs().max()) + 1e-12


def test_part_mlp_hip_backward_vs_torch_autograd(gpu_setup):
    """Row f1: invr_part_mlp_fwd / invr_part_mlp_bwd (PartMlpFn) against the same two MLPs as torch ops under
    torch.autograd: outputs, the embedding gradient and every parameter gradient (both colour-net depths, ragged n)."""
    from invr import autograd as AG
    cfg, sd, batch, gb, net = gpu_setup
    g = torch.Generator().manual_seed(5)
    keep = []
    model = _abi.make_model({k: v for k, v in net.named_parameters()}, cfg, keep)
    n_freq = cfg.viewdir_embedder.kwargs['res']
    for pid, n in ((0, 5000), (1, 1234), (2, 17), (4, 2048)):
        pn = net.tpose_human.part_networks[pid]
        emb0 = torch.randn(n, 19, generator=g) * 0.5
        dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1) * (0.5 + torch.rand(n, 1, generator=g))
        gr = torch.randn(n, 4, generator=g)
        params = [p for m in (pn.occ, pn.rgb) for l in m.linears for p in (l.weight, l.bias)] + [pn.rgb_latent]
        res = []
        for hip in (False, True):
            emb = cu(emb0).requires_grad_()
            for p in params:
                p.grad = None
            if hip:
                plist = [p for m in (pn.occ, pn.rgb) for l in m.linears for p in (l.weight, l.bias)]
                raw = AG.PartMlpFn.apply(emb, cu(dirs), model, pid, gb['latent_index'], pn.rgb_latent, *plist)
            else:
                raw = AG.part_mlps_torch(pn, emb, cu(dirs), gb['latent_index'], n_freq)
            (raw * cu(gr)).sum().backward()
            res.append([raw.detach(), emb.grad] + [p.grad.clone() for p in params])
        names = ['raw', 'g_emb'] + ['param%d' % k for k in range(len(params))]
        for nm, a, b in zip(names, *res):
            scale = float(a.abs().max()) + 1e-12
            assert maxerr(a, b) <= 2e-5 * scale + 1e-7, (pid, n, nm, maxerr(a, b), scale)
    for p in net.parameters():
        p.grad = None


def test_renderer_adaptive_workspace_capacity(gpu_setup, golden):
    """Renderer sizes the survivor capacity from the previous frame; an undersized guess is detected through
    stats[6] and the frame rendered again — outputs are identical either way."""
    cfg, sd, batch, gb, net = gpu_setup
    r = Renderer(net)
    a = r.render(dict(gb))
    assert r._cap_hint is not None and r._cap_hint >= int(r.last_stats[0])
    r._cap_hint = 10                                        # far too small -> overflow -> full-capacity retry
    b = r.render(dict(gb))
    r.adaptive_cap = False
    c = r.render(dict(gb))
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert int(r.last_stats[6]) == 0


def test_row_sum_tables_follow_weight_updates(small_setup):
    """The eval-only row-sum tables are derived data: after the tables change (in-place copy, FusedAdam step through
    raw pointers) the next eval render must rebuild them."""
    from invr.optim import FusedAdam
    cfg, sd, batch, extras = small_setup
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    gb = {k: v.to(DEV) for k, v in batch.items()}
    r = Renderer(net)
    a = r.render(dict(gb))['rgb_map']
    emb = net.tpose_human.part_networks[0].embedder
    opt = FusedAdam([{'params': [emb.hash, emb.dense], 'lr': 0.1}], 0.1, eps=1e-15)
    g = torch.Generator(device=DEV).manual_seed(3)
    emb.hash.grad = torch.randn(emb.hash.shape, generator=g, device=DEV)
    emb.dense.grad = torch.randn(emb.dense.shape, generator=g, device=DEV)
    opt.step()
    b = r.render(dict(gb))['rgb_map']
    with encoder_mode(cfg, False):                           # 64-byte rows: no derived table involved
        c = r.render(dict(gb))['rgb_map']
    assert float((a - b).abs().max()) > 1e-3                 # the update is visible
    assert float((b - c).abs().max()) < 1e-4                 # and the row-sum path agrees with the direct path


@pytest.mark.parametrize('scene_kw', [dict(seed=1, frame=17, pose_scale=0.9), dict(seed=2, frame=60, cam_dist=1.6),
                                      dict(seed=5, frame=99, pose_scale=0.2, cam_dist=4.0)])
def test_render_other_scenes_vs_oracle(scene_kw):
    """Other bodies / poses / frames / camera distances than the golden scene (different band / far-pair geometry,
    latent code and deformer time slice), against the reference-pinned oracle; float64-arbitrated like the variants."""
    from invr import scene
    from invr.config import make_cfg
    cfg = make_cfg(table_log2=12, N_samples=32)
    sd = params.init_state_dict(cfg, seed=21 + scene_kw['seed'])
    batch_np, _ = scene.make_scene(48, 48, **scene_kw)
    batch = scene.to_torch(batch_np)
    sel = torch.arange(0, batch['ray_o'].shape[1], 5)
    b = dict(batch)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = batch[k][:, sel]
    net = Network(cfg=cfg)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    r = Renderer(net)
    ret = r.render({k: v.to(DEV) for k, v in b.items()})
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        exact = O.render(O.Model(sd64, cfg), b64)['rgb_map'][0]
    assert ((ret['raw'][0, :, 3] != 0) == (ref['raw'][0, :, 3] != 0)).all()
    assert int((ref['raw'][0, :, 3] != 0).sum()) > 200
    err_gpu = (ret['rgb_map'][0].double() - exact).abs().max(1)[0]
    err_ref = pixel_noise(O, O.Model(sd64, cfg), b64, exact, int(cfg.N_samples), chunk=4096, ref32=ref['rgb_map'][0], trials=4)      # tests/conditioning.py
    assert bool((err_gpu <= 1e-4 + 4 * err_ref).all()), (float(err_gpu.max()), float(err_ref.max()))
    assert float(err_gpu.median()) < 5e-6


@pytest.mark.parametrize('scale', [1.0, 40.0, float('nan')], ids=['as-is', 'x40', 'nan'])
def test_far_fold_distance_follows_the_frame_matrices(gpu_setup, scale):
    """Far-constant folding (k_knn.hip header) rests on |A_bw| <= s * max|A| being below fp32 resolution; the distance beyond
    which a (survivor, part) pair is folded is derived per frame from the largest |entry| of A / big_A (k_part_prepare ->
    ix.dfar2): (0.68 m)^2 while that is <= 2, larger beyond, +inf (no folding) for non-finite matrices.  What the fold claims is
    then checked directly: the dense warp (invr_knn_blend + invr_warp_deform, the reference's arithmetic for every pair) puts
    every folded pair's canonical point within 1.5e-8 m of the origin with a view direction below 1e-15."""
    from invr import stages
    cfg, sd, batch, gb, net = gpu_setup
    b = dict(gb)
    b['A'] = gb['A'] * scale
    b['big_A'] = gb['big_A'] * scale
    ctx = net.prepare(b)
    ro, rd, nr, fa = (b[k][0] for k in ('ray_o', 'ray_d', 'near', 'far'))
    S = int(cfg.N_samples)
    net._ws = None
    out = net.render_rays(ctx, ro, rd, nr, fa, S, want_raw=False)
    torch.cuda.synchronize()
    net._ws = None
    v = _abi.ws_views(*out['_ws'])
    st = out['stats'].cpu().numpy()
    dfar2 = float(v['knn_dfar2'][0])
    if scale != scale:
        assert dfar2 == float('inf') and int(st[7:12].sum()) == 0 and int(v['farflags'][:int(st[0])].max()) == 0
        return
    M = max(float(b['A'].abs().max()), float(b['big_A'].abs().max()))
    if M <= 2.0:
        assert dfar2 == float(np.float32(0.4624)), (M, dfar2)
    else:
        assert dfar2 > 0.4624 and abs(dfar2 - 0.01125 * np.log(4.0 * M / 1.12e-17)) < 1e-4, (M, dfar2)
    Na = int(st[0])
    assert st[6] == 0 and Na > 1000 and int(st[7:12].sum()) > 100
    act = v['active_idx'][:Na]
    pts, dirs = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, act)
    far = ((v['farflags'][:Na, None].to(torch.int32) >> torch.arange(5, device=DEV)[None]) & 1).bool()
    nn, d2, w, dist = stages.knn_neighbors(ctx.scene, pts)
    assert bool((d2[..., 0][far] > dfar2 * 0.9999).all())                      # folded = nearest vertex beyond the far distance
    bw, _ = stages.knn_blend(ctx.scene, pts)
    tp, td, rs = stages.warp_deform(ctx.scene, ctx.model, pts, dirs, bw, far)
    x0 = (tp - rs)[far]                                                          # init_bigpose of the folded pairs
    assert float(x0.abs().max()) <= 1.5e-8, float(x0.abs().max())
    assert float(td[far].abs().max()) <= 1e-15, float(td[far].abs().max())


def test_training_backward_refuses_an_overwritten_workspace(gpu_setup, golden):
    """The fused training forward keeps its pair lists / activations in the network's shared workspace.  A second library call
    on the network before the backward (gradient accumulation over two forwards, an eval render) overwrites them: the backward
    and the lazy train-mode tensors must raise instead of differentiating the other call's lists."""
    import copy
    from invr.trainer import NetworkWrapper
    cfg, sd, batch, gb, net0 = gpu_setup
    net = copy.deepcopy(net0).train()
    net.cfg = copy.deepcopy(cfg)
    wrap = NetworkWrapper(net)
    tb = _train_batch(gb, golden)
    tb['iter_step'] = 2
    ret1, loss1, _, _ = wrap(dict(tb), split='train')
    ret2, loss2, _, _ = wrap(dict(tb), split='train')           # second forward: takes over the workspace
    with pytest.raises(RuntimeError, match='workspace'):
        loss1.backward()
    with pytest.raises(RuntimeError, match='workspace'):
        ret1['tocc']
    assert ret2['tocc'].shape[1] == ret2['resd'].shape[1]         # the latest forward is fine
    loss2.backward()
    assert sum(float(p.grad.abs().sum()) for p in net.parameters() if p.grad is not None) > 0
