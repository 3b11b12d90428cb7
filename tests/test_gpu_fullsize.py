"""GPU, BASELINE full sizes (inb_377 defaults: 285,993,711 parameters, 512x512 frame, 128 samples):
size-independent properties of the render + a spot check against the oracle on a ray subset,
and the edge cases (empty / ragged ray lists, S not a multiple of 64, nothing surviving the cull)."""
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


from oracle import nvr_oracle as O          # noqa: E402  (checker only)
from invr import scene                      # noqa: E402
from invr.config import make_cfg            # noqa: E402
from invr.network import Network            # noqa: E402

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def full(full_net):
    cfg, net = full_net
    bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8)
    bc = scene.to_torch(bnp)
    gb = {k: v.to(DEV) for k, v in bc.items()}
    return cfg, net, bc, gb


def rays(gb, idx=None):
    r = [gb[k][0] for k in ('ray_o', 'ray_d', 'near', 'far')]
    return r if idx is None else [t[idx] for t in r]


def test_full_frame_properties(full):
    cfg, net, bc, gb = full
    ctx = net.prepare(gb)
    n = gb['ray_o'].shape[1]
    assert n > 250000
    a = net.render_rays(ctx, *rays(gb), 128, want_raw=False)
    rgb = a['rgb_map'].clone(); acc = a['acc_map'].clone(); st = a['stats'].cpu().numpy()
    assert st[6] == 0 and 0 < st[0] < n * 128
    assert bool(torch.isfinite(rgb).all()) and float(rgb.min()) >= 0 and float(rgb.max()) <= 1
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5
    # determinism: same inputs -> bit-identical outputs (list order is not deterministic, results are)
    b = net.render_rays(ctx, *rays(gb), 128, want_raw=False)
    assert torch.equal(b['rgb_map'], rgb) and torch.equal(b['acc_map'], acc)
    # rays are independent: any split / permutation of the ray list gives the same pixels
    perm = torch.randperm(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    c = net.render_rays(ctx, *rays(gb, perm), 128, want_raw=False)
    assert torch.equal(c['rgb_map'], rgb[perm]) and torch.equal(c['acc_map'], acc[perm])
    for sl in (slice(0, 100000), slice(100000, n)):
        d = net.render_rays(ctx, *rays(gb, torch.arange(n, device=DEV)[sl]), 128, want_raw=False)
        assert torch.equal(d['rgb_map'], rgb[sl])
    # compositing is a convex combination: rgb_map <= acc_map * max rgb
    assert bool((rgb.max(1)[0] <= acc + 1e-5).all())
    # nothing reads workspace memory it has not written this frame: fresh workspaces over dirtied allocator blocks give the same
    # bits (the lattice-cell records, pair lists, flag bytes and counters all live in the caller's uninitialised buffer)
    sel = torch.randperm(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))[:4096]
    ref = None
    for it in range(6):
        net._ws = None
        junk = torch.randint(0, 255, (int(6e8),), dtype=torch.uint8, device=DEV)
        del junk
        o = net.render_rays(ctx, *rays(gb, sel), 128, want_raw=True)
        cur = (o['rgb_map'].clone(), o['raw'].clone(), o['stats'].clone())
        if ref is None:
            ref = cur
        assert all(torch.equal(x, y) for x, y in zip(ref, cur)), it


def test_full_size_spot_check_vs_oracle(full):
    """64 rays x 128 samples with the full 1.09 GB (random, untrained) tables against the CPU oracle.
    Survivor / flag decisions must be identical.  Values: <= 1e-4 per pixel, except where the
    reference arithmetic itself is ill-conditioned: band/far pairs land outside the part boxes and are
    EXTRAPOLATED with trilinear weights of 1e3..1e6, which amplifies fp32 rounding in
    the reference as much as in the kernels.  That is measured, not assumed: the oracle is also run
    in float64, and a pixel may deviate from the float64 result by 1e-4 + 4x the deviation of the
    reference's own float32 arithmetic on that pixel."""
    cfg, net, bc, gb = full
    n = gb['ray_o'].shape[1]
    sel = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:64].sort()[0]
    out = net.render_rays(gb, *rays(gb, sel.to(DEV)), 128, want_raw=True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    b = dict(bc)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = bc[k][:, sel]
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b, n_samples=128)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        ref64 = O.render(O.Model(sd64, cfg), b64, n_samples=128)
    assert int((ref['occ'][0, :, 0] != 0).sum()) > 100
    nz_ref = ref['raw'][0, :, 3] != 0
    nz = out['raw'].cpu()[:, 3] != 0
    assert bool((nz == nz_ref).all())                        # identical survivor / flag decisions
    exact = ref64['rgb_map'][0]
    err_gpu = (out['rgb_map'].cpu().double() - exact).abs().max(1)[0]
    # noise scale of a pixel: tests/conditioning.py (one fp32 run alone is a single noisy sample of the conditioning)
    from tests.conditioning import pixel_noise
    err_ref = pixel_noise(O, O.Model(sd64, cfg), b64, exact, 128, ref32=ref['rgb_map'][0],
                          rerun32=lambda ch: O.render(O.Model(sd, cfg), b, n_samples=128, chunk=ch)['rgb_map'][0])
    _tol_report('spot check: needed multiple of the noise scale max((err - 1e-4) / noise) = %.3f' % float(((err_gpu - 1e-4) / err_ref.clamp(min=1e-12)).max()))
    assert bool((err_gpu <= 1e-4 + 4 * err_ref).all()), (float(err_gpu.max()), float(err_ref.max()))        # (4 x; five runs of round 6 needed none: gpurun_out/r6tol)
    assert float(err_gpu.median()) < 2e-6
    well = err_ref < 2e-6                                    # well-conditioned pixels: plain fp32 bar
    assert int(well.sum()) >= 32 and float(err_gpu[well].max()) < 1e-4


def test_edge_cases(full):
    cfg, net, bc, gb = full
    ctx = net.prepare(gb)
    e = torch.empty(0, 3, device=DEV)
    z = net.render_rays(ctx, e, e, torch.empty(0, device=DEV), torch.empty(0, device=DEV), 128)
    assert z['rgb_map'].shape == (0, 3) and z['acc_map'].shape == (0,)
    idx = torch.arange(0, 999, device=DEV) * 251                        # ragged: 999 rays
    for S in (2, 3, 63, 65, 200):                                        # S not a multiple of the wave size
        o = net.render_rays(ctx, *rays(gb, idx), S, want_raw=True)
        assert o['rgb_map'].shape == (999, 3) and o['raw'].shape == (999 * S, 4)
        assert bool(torch.isfinite(o['rgb_map']).all())
    # nothing survives the cull: rays far away from the body -> exact zeros, no pairs
    ro = gb['ray_o'][0][:500] + 50.0
    o = net.render_rays(ctx, ro, gb['ray_d'][0][:500], gb['near'][0][:500], gb['far'][0][:500], 64, want_raw=True)
    assert int(o['stats'][0]) == 0 and float(o['rgb_map'].abs().max()) == 0 and float(o['raw'].abs().max()) == 0
    # a too-small max_active is reported, not silently wrong
    o = net.render_rays(ctx, *rays(gb, torch.arange(20000, device=DEV)), 128, max_active=1000)
    assert int(o['stats'][6]) == 1


def test_generate_rays_full_frame(full):
    """512x512: device ray generation vs the NumPy restatement of the reference's get_rays_within_bounds
    (itself asserted bit-equal to the reference in tests/golden/make_golden_rays.py)."""
    from invr import rays
    cfg, net, bc, gb = full
    _, ex = scene.make_scene(512, 512, seed=0, cam_dist=1.8)
    ro, rd, near, far, mask = rays.rays_within_bounds(512, 512, ex['K'], ex['Rc'], ex['Tc'], bc['wbounds'][0].numpy(), DEV)
    assert np.array_equal(mask.cpu().numpy().reshape(-1), bc['mask_at_box'][0].numpy())      # byte result: exact
    assert torch.equal(ro.cpu(), bc['ray_o'][0])
    for a, b in ((rd, bc['ray_d'][0]), (near, bc['near'][0]), (far, bc['far'][0])):
        a = a.cpu()
        assert a.shape == b.shape
        # float64 products may be fused differently by the host BLAS: allow a handful of 1-ulp float32 flips
        bad = (a != b)
        assert int(bad.sum()) <= 8 and float((a - b).abs().max()) <= 3e-7


def test_reference_style_init_every_pixel_within_1e4():
    """The tables of the other full-size tests are N(0, 0.1^2): ~300x larger than what the reference's own initialisation
    gives (part_base_embedder.py:72,79: ONE kaiming_normal_ over the (levels, T, 16) tensor, fan_in = T * 16 -> std 3.4e-4
    for the 2^20-row grids), which is what makes the reference's extrapolation outside a part's box ill-conditioned there.  Here
    the model is initialised exactly as the reference does (Network() = the same constructor calls, MLPs at torch's default
    init), at full size (285,993,711 parameters), and the PLAIN bar applies to everything: 4135 rays x 128 samples of the bench
    frame, every pixel of rgb_map / acc_map and every sample of raw within 1e-4 of the oracle, survivor set identical."""
    import copy
    from oracle import nvr_oracle as O          # checker only
    from invr import scene
    from invr.config import make_cfg
    from invr.network import Network
    torch.manual_seed(20260927)
    cfg = make_cfg(N_samples=128)
    with torch.device(DEV):
        net = Network(cfg=copy.deepcopy(cfg))
    net = net.to(DEV).eval()
    assert sum(p.numel() for p in net.parameters()) == 285993711
    e = net.tpose_human.part_networks[0].embedder
    std = float(e.hash.detach().std())
    assert 2.5e-4 < std < 4.5e-4, std                               # sqrt(2 / (1048583 * 16)) = 3.45e-4
    bnp, _ = scene.make_scene(512, 512, seed=0, cam_dist=1.8)
    bc = scene.to_torch(bnp)
    n = bc['ray_o'].shape[1]
    sel = torch.arange(7, n, 62)
    assert sel.numel() >= 4096
    b = dict(bc)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = bc[k][:, sel]
    gb = {k: v.to(DEV) for k, v in b.items()}
    ctx = net.prepare(gb)
    out = net.render_rays(ctx, gb['ray_o'][0], gb['ray_d'][0], gb['near'][0], gb['far'][0], 128, want_raw=True)
    torch.cuda.synchronize()
    st = out['stats'].cpu().numpy()
    assert st[6] == 0 and st[0] > 20000
    sd = {k: t.detach().cpu() for k, t in net.state_dict().items()}
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b, n_samples=128, chunk=512)
    raw, rref = out['raw'].cpu(), ref['raw'][0]
    assert bool(((raw[:, 3] != 0) == (rref[:, 3] != 0)).all())      # identical survivor / flag decisions
    e_rgb = (out['rgb_map'].cpu() - ref['rgb_map'][0]).abs().max(1)[0]
    e_acc = (out['acc_map'].cpu() - ref['acc_map'][0]).abs()
    e_raw = (raw - rref).abs().max(1)[0]
    print('reference-style init: %d rays, %d survivors; max |d rgb_map| %.2e, |d acc_map| %.2e, |d raw| %.2e'
          % (sel.numel(), int(st[0]), float(e_rgb.max()), float(e_acc.max()), float(e_raw.max())))
    assert float(e_rgb.max()) <= 1e-4 and float(e_acc.max()) <= 1e-4          # every pixel, no allowance
    assert float(e_raw.max()) <= 1e-4                                         # every ray-sample
    assert float(ref['rgb_map'][0].abs().max()) > 0.05                        # (the image is not trivially black)


def order_frames(tmp_path, dev, res, S, extra=()):
    """The same seeded frame rendered by two processes, ray-major (INVR_ORDER=0) and depth-windowed (default) survivor order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for order in ('0', '1'):
        out = str(tmp_path / ('order%s.npz' % order))
        r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'order_frame.py'), out, dev, str(res), str(S)] + list(extra),
                           env=dict(os.environ, INVR_ORDER=order), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(dict(np.load(out)))
    return outs


def check_more_sample_counts(a, b, counts):
    """every power-of-two sample count the windowed order takes (8 ... 1024: 1 ... 128 windows per ray, 1024 ... 8 rays per block)"""
    for S2 in counts:
        act0, act1 = a['act_%d' % S2].astype(np.int64), b['act_%d' % S2].astype(np.int64)
        assert act0.size > 200 and np.array_equal(act0, np.sort(act0)) and np.array_equal(np.sort(act1), act0), S2
        rows = 8192 // S2
        ray, smp = act1 // S2, act1 % S2
        key = ((ray // rows) * (S2 // 8) + smp // 8) * 8192 + (ray % rows) * 8 + smp % 8
        assert bool((key[1:] > key[:-1]).all()), S2
        assert np.array_equal(a['stats_%d' % S2][:12], b['stats_%d' % S2][:12]), S2
        for k in ('rgb', 'occ'):
            assert np.array_equal(a['%s_%d' % (k, S2)], b['%s_%d' % (k, S2)]), (k, S2)


def check_order_frames(a, b, S, min_survivors):
    na = int(a['stats'][0])
    assert na > min_survivors and na == int(b['stats'][0]) and a['stats'][6] == 0 and b['stats'][6] == 0
    assert np.array_equal(a['stats'][:12], b['stats'][:12])                  # survivor, pair and far-pair counts per part
    # ray-major: ascending; windowed: ranked by (8-sample window, ray, sample) inside blocks of 8192 / S rays — and NOT ascending
    act0, act1 = a['act'].astype(np.int64), b['act'].astype(np.int64)
    assert np.array_equal(act0, np.sort(act0)) and np.array_equal(np.sort(act1), act0) and not np.array_equal(act1, act0)
    rows = 8192 // S
    ray, smp = act1 // S, act1 % S
    key = ((ray // rows) * (S // 8) + smp // 8) * 8192 + (ray % rows) * 8 + smp % 8
    assert bool((key[1:] > key[:-1]).all())
    for k in ('rgb', 'acc', 'raw', 'occ'):                                   # nothing downstream depends on the slot order: bit for bit
        assert np.array_equal(a[k], b[k]), k


def test_survivor_order_does_not_change_the_frame(tmp_path):
    """Eval frames rank their survivors by depth window inside blocks of 64 rays (csrc/k_cull.hip) so that a KNN ticket /
    an encoder wave is one compact slab; the pair lists, the merge and the compositing only see slots.  The 512x512x128 bench frame
    (2^12-row tables, seeded) rendered in both orders: same survivors, same counts, rgb_map / acc_map / raw / occ bit-identical."""
    a, b = order_frames(tmp_path, DEV, 512, '128,8,16,32,64,256')
    check_order_frames(a, b, 128, 1000000)
    check_more_sample_counts(a, b, (8, 16, 32, 64, 256))
