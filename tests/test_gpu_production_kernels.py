"""GPU, BASELINE full size: the kernels the frame actually spends its time in, pinned DIRECTLY.

`invr_render_fwd` runs `k_knn_pairs` (cluster / sub-cluster / lattice-cell pruned 4-NN with placeholder-seeded
top-4), `k_warp_pairs` + `k_deform_pairs_slice` (pre-blended per-vertex matrices, MFMA deformer on per-frame
t-slices) and `k_part_encode_rs_xcd` (row-sum tables, XCD-partitioned level groups).  The stage entry points of
include/invr.h run other kernels (brute force / dense / generic), which tests/test_gpu_parity.py pins to the
reference goldens.  Here the production kernels' own per-pair results are read out of the workspace
(`_abi.ws_views`: l_slot, l_nn, l_w, pflags, farflags, l_x, l_d, l_r, emb) for the WHOLE 512x512x128 frame
(1.6-7 M survivors x 5 parts) at four poses, including pose_scale >= 1.0 and smpl_thresh 0.1 (inb_lan.yaml), and
compared with

  * the brute-force KNN (`invr_knn_neighbors`) on the identical pose points (`invr_pose_points`): flag / far
    decisions and neighbour rows bit-exact (ties are ordered by the (distance,row) key), weights bit-exact,
  * the oracle's `knn_blend` + `dist < thresh` (blend_utils.py:732-763,817-825) on a random subset,
  * the dense warp + point deformer (`invr_warp_deform`) for every listed pair,
  * the oracle's `hash_embed` (part_base_embedder.py:106-174) with the real 1.09 GB tables on >= 1e5 pairs,
  * the whole render against the oracle with the strict 1e-4 bar on every well-conditioned pixel.
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


from oracle import nvr_oracle as O          # noqa: E402  (checker only)
from invr import _abi, scene, stages        # noqa: E402

DEV = 'cuda:0'
RES, S = 512, 128
# least counts a frame must reach for the comparisons to mean something, and sample sizes (tests/test_hostsim_cpu.py runs the same
# test bodies on a reduced frame with smaller numbers)
MIN = dict(na=500000, listed=500000, oracle_subset=6000, oracle_chunk=2000, enc_take=25000, enc_total=100000, enc_inside=30000,
           strict_rays=256, occ=300)
WELL_FLOOR = 0.90        # least fraction of the 256 sampled pixels per pose that must be well conditioned and meet the plain 1e-4 bar
                         # (measured: 240-256 of 256 at the four poses with the multi-trial noise estimate of tests/conditioning.py; printed)
POSES = [dict(seed=0, pose_scale=0.5, frame=3, cam_dist=1.8, thresh=0.05),
         dict(seed=1, pose_scale=1.0, frame=17, cam_dist=1.8, thresh=0.05),
         dict(seed=2, pose_scale=1.2, frame=60, cam_dist=2.2, thresh=0.1),          # inb_lan.yaml smpl_thresh
         dict(seed=3, pose_scale=0.8, frame=99, cam_dist=2.6, thresh=0.05),
         dict(seed=0, pose_scale=0.5, frame=3, cam_dist=1.8, thresh=0.1)]           # the bench frame at mid density: 15 % of the ray-samples survive (7 % at 0.05)

@pytest.fixture(scope='module', params=range(len(POSES)), ids=['pose%d' % i for i in range(len(POSES))])
def fr(request, full_net):
    """One production render of a pose (all rays, 128 samples, full 28 GB workspace) and views of what it left behind;
    module-scoped and parametrised, so pytest runs every test of a pose on one render."""
    return make_frame(request.param, *full_net)


def make_frame(k, cfg0, net):
    kw = dict(POSES[k])
    thresh = kw.pop('thresh')
    bnp, _ = scene.make_scene(RES, RES, **kw)
    bc = scene.to_torch(bnp)
    gb = {k_: v.to(DEV) for k_, v in bc.items()}
    cfg = copy.deepcopy(cfg0)
    cfg.smpl_thresh = thresh
    old = net.cfg
    net.cfg = cfg
    try:
        ctx = net.prepare(gb)
        ro, rd, nr, fa = (gb[k_][0] for k_ in ('ray_o', 'ray_d', 'near', 'far'))
        net._ws = None                                  # own workspace: the views must stay valid while cached
        out = net.render_rays(ctx, ro, rd, nr, fa, S, want_raw=False)
        torch.cuda.synchronize()
        net._ws = None
    finally:
        net.cfg = old
    st = out['stats'].cpu().numpy()
    assert st[6] == 0
    v = _abi.ws_views(*out['_ws'])
    Na = int(st[0])
    act = v['active_idx'][:Na]
    pts, dirs = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, act)
    return dict(k=k, cfg=cfg, net=net, bc=bc, gb=gb, ctx=ctx, out=out, st=st, v=v, Na=Na, act=act, pts=pts, dirs=dirs, thresh=thresh)


def ulp_diff(a, b):
    """|a - b| in units in the last place (float32, same-sign finite values)."""
    return (a.view(torch.int32).long() - b.view(torch.int32).long()).abs()


def test_cull_survivor_set_exact_whole_frame(fr):
    """The near-surface cull of the frame path (k_front_cull: cell mask + the one-multiply pre-test, csrc/front_bodies.h) keeps EXACTLY
    the samples whose pose-space distance — the dense stage kernels, the reference's arithmetic bit for bit (inb_renderer.py:15-31 +
    blend_utils.py sample of the distance channel, :135's `pnorm < smpl_thresh`) — is below the threshold: every ray-sample of the
    frame, no sample more, none less."""
    from invr.autograd import sample_volume
    f = fr
    gb, thresh = f['gb'], f['thresh']
    ro, rd, nr, fa = (gb[k_][0] for k_ in ('ray_o', 'ray_d', 'near', 'far'))
    n = ro.shape[0]
    vol, bounds = gb['pbw'][0].contiguous(), gb['pbounds'][0].contiguous()
    keep = torch.zeros(n * S, dtype=torch.bool, device=ro.device)
    step = max(1, (1 << 22) // S)                                          # rays per slab (4 M samples)
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        pts, _ = stages.pose_points(f['ctx'].scene, ro[r0:r1], rd[r0:r1], nr[r0:r1], fa[r0:r1], S, want_dirs=False)
        pn = sample_volume(vol, bounds, pts, vol.shape[3] - 1, 1)[:, 0]
        keep[r0 * S:r1 * S] = pn < thresh
    want = keep.nonzero(as_tuple=True)[0]
    got = f['act'].long().sort()[0]
    assert want.numel() == f['Na'] and torch.equal(got, want), (want.numel(), f['Na'])


def test_knn_pairs_vs_brute_force_whole_frame(fr):
    f, k = fr, fr['k']
    v, st, Na, thresh = f['v'], f['st'], f['Na'], f['thresh']
    assert Na > MIN['na']
    # survivors: every sample once — in ray-major order, or (eval frames with a power-of-two sample count, csrc/k_cull.hip "windowed
    # survivor order") ranked by (8-sample depth window, ray, sample) inside blocks of 8192 ray-samples
    act = f['act'].long()
    asc = act.sort()[0]
    assert bool((asc[1:] > asc[:-1]).all())
    if not torch.equal(act, asc):
        assert S >= 8 and S & (S - 1) == 0
        rows = 8192 // S
        ray, smp = act // S, act % S
        key = ((ray // rows) * (S // 8) + smp // 8) * 8192 + (ray % rows) * 8 + smp % 8
        assert bool((key[1:] > key[:-1]).all())
    # the (ray-sample -> slot) map the compositing uses: rank of the sample's mask word / byte + popcount of the bits below it
    tab = torch.tensor([bin(x).count('1') for x in range(256)], device=DEV)
    word, bit = v['mask'][act >> 6], act & 63
    below = word & ((torch.ones_like(bit) << bit) - 1)
    slot_w = v['word_off'][act >> 6].long() + sum(tab[(below >> (8 * k)) & 255] for k in range(8))               # ray-major calls
    slot_b = v['byte_off'][act >> 3].long() + tab[((word >> (bit & ~7)) & 255) & ((torch.ones_like(bit) << (bit & 7)) - 1)]     # windowed calls
    ar = torch.arange(Na, device=DEV)
    assert torch.equal(slot_b, ar) if not torch.equal(act, asc) else (torch.equal(slot_w, ar) or torch.equal(slot_b, ar))
    nn, d2, w, dist = stages.knn_neighbors(f['ctx'].scene, f['pts'])
    pf, ff = v['pflags'][:Na].int(), v['farflags'][:Na].int()
    assert int((pf & ff).max()) == 0
    n_listed = 0
    for p in range(5):
        listed, far = ((pf >> p) & 1).bool(), ((ff >> p) & 1).bool()
        ref_flag = dist[:, p] < thresh                                           # inb_part_network_multiassign.py:90
        # the reference flags near pairs AND far pairs (epsilon-normalised weights): listed | far must be
        # exactly that set, for every survivor of the frame
        assert torch.equal(listed | far, ref_flag), (k, p, int(((listed | far) != ref_flag).sum()))
        # far pairs are replaced by the part constant: legal only beyond 0.68 m from the part's nearest vertex
        if bool(far.any()):
            assert float(d2[far, p, 0].min()) > 0.4624
        cnt = int(st[1 + p]) - 1                                                 # last entry = the far-constant pair
        slots = v['l_slot'][p][:cnt].long()
        cap = v['cap']                                                           # neighbours / weights live at the survivor's slot
        assert int(v['l_slot'][p][cnt]) == cap and int(v['l_nn'][p][cap].abs().max()) == 0 and float(v['l_w'][p][cap].abs().max()) == 0
        assert cnt == int(listed.sum())
        assert torch.equal(slots, listed.nonzero(as_tuple=True)[0])              # every listed pair exactly once, in ascending slot order
        # neighbour rows: bit-exact, in (distance,row) order; weights: same arithmetic on the same distances -> bit-exact
        assert torch.equal(v['l_nn'][p][slots], nn[slots, p]), (k, p)
        wd = ulp_diff(v['l_w'][p][slots], w[slots, p])
        assert int(wd.max()) <= 1, (k, p, int(wd.max()))
        n_listed += cnt
    assert n_listed > MIN['listed']
    # brute-force kernel vs the oracle (CPU restatement of pytorch3d knn_points + sample_blend_closest_points) on a subset
    sel = torch.randperm(Na, generator=torch.Generator().manual_seed(k))[:MIN['oracle_subset']].to(DEV)
    b = f['bc']
    pp = f['pts'][sel].cpu()
    got_nn, got_dist, got_w = nn[sel].cpu(), dist[sel].cpu(), w[sel].cpu()
    for c0 in range(0, sel.numel(), MIN['oracle_chunk']):
        sl = slice(c0, c0 + MIN['oracle_chunk'])
        P, M = b['part_pts'][0].shape[:2]
        dd = ((pp[sl][None, :, None, :] - b['part_pts'][0][:, None, :, :]) ** 2).sum(-1)
        dd = dd.masked_fill(torch.arange(M)[None, None, :] >= b['lengths2'][0][:, None, None], float('inf'))
        d2k, idx = dd.topk(4, dim=-1, largest=False)                             # (P,n,4)
        # neighbour SETS (knn_points order is unspecified).  The CPU sum may associate (dx2+dy2)+dz2 differently, so a 4th/5th
        # neighbour at (near-)equal distance may swap: the kernel's four rows must have the oracle's four smallest distances
        ref_sets = idx.permute(1, 0, 2).sort(-1)[0]
        same = (got_nn[sl].long().sort(-1)[0] == ref_sets).all(-1)               # (n,P)
        assert float(same.float().mean()) > 0.999
        got_d = torch.gather(dd.permute(1, 0, 2), 2, got_nn[sl].long()).sort(-1)[0]
        ref_d = d2k.permute(1, 0, 2).sort(-1)[0]
        fin = torch.isfinite(ref_d)
        assert bool((fin == torch.isfinite(got_d)).all())
        assert bool(((got_d - ref_d).abs()[fin] <= 1e-6 * ref_d[fin] + 1e-12).all()), k
        _, ref_dist = O.knn_blend(pp[sl], b['part_pts'][0], b['part_pbw'][0], b['lengths2'][0])
        assert float((got_dist[sl] - ref_dist).abs().max()) < 2e-6
        margin = (ref_dist - thresh).abs() < 1e-6
        assert bool((((got_dist[sl] < thresh) == (ref_dist < thresh)) | margin).all())


def test_warp_pairs_and_slice_deformer_vs_dense_whole_frame(fr):
    f, k = fr, fr['k']
    v, st, Na = f['v'], f['st'], f['Na']
    ctx = f['ctx']
    bw, _ = stages.knn_blend(ctx.scene, f['pts'])
    pf = v['pflags'][:Na].int()
    flag = torch.stack([((pf >> p) & 1) for p in range(5)], 1).to(torch.uint8)
    tp, td, rs = stages.warp_deform(ctx.scene, ctx.model, f['pts'], f['dirs'], bw, flag)
    del bw
    worst = {}
    for p in range(5):
        cnt = int(st[1 + p]) - 1
        if cnt == 0:
            continue
        slots = v['l_slot'][p][:cnt].long()
        r = v['l_r'][p][:, :cnt].t().contiguous()
        xb = (v['l_x'][p][:, :cnt].t() - r).contiguous()                           # init_bigpose (l_x = init_bigpose + resd)
        d = v['l_d'][p][:, :cnt].t()
        ex = (xb - (tp[slots, p] - rs[slots, p])).abs().max(1)[0]
        ed = (d - td[slots, p]).abs().max(1)[0]
        # residual: 0.05 tanh(MLP(grid(uv(x)))) is steep in x (nearest-vertex UV volume: d uv / d x ~ 40 / m), so it is
        # compared on the SAME canonical point: MFMA + per-frame t-slices (k_deform_pairs_slice) vs the thread-per-point
        # deformer with 3-D lookups (invr_deform_fwd, pinned to the reference goldens by test_warp_deform)
        r_pts = f['net'].resd(xb[None], ctx)[0]
        er = (r - r_pts).abs().max(1)[0]
        worst[p] = (float(ex.max()), float(ed.max()), float(er.max()))
        # canonical point / view direction: the pre-blended per-vertex matrices change the summation order of the
        # 24-joint blend (sum_k w_k (sum_j pbw_kj A_j) vs (sum_k w_k pbw_kj) A_j): fp32 rounding of O(1) quantities
        assert float(ex.max()) < 1e-5 and float(ed.max()) < 1e-5, (k, p, worst[p])
        # (xb is recovered as l_x - l_r, one rounding away from what the kernel saw: where the nearest-vertex UV volume jumps the
        # residual moves by up to ~1e-4 for that ulp, so the bound is on all but a 1e-4 fraction of the ~1e6 pairs)
        assert float((er > 2e-6).float().mean()) < 1e-4 and float(er.max()) < 1e-3, (k, p, worst[p], float((er > 2e-6).float().mean()))
        assert float(r.abs().max()) <= 0.05 + 1e-7
        # and the dense path's residual of ITS canonical point agrees within the deformer's sensitivity to that fp32 noise
        assert float((r - rs[slots, p]).abs().max()) < 2e-3, (k, p)
    assert len(worst) >= 4
    # the far-constant pair: zero weights -> canonical origin, zero direction (k_knn.hip header)
    for p in range(5):
        c = int(st[1 + p]) - 1
        x0 = v['l_x'][p][:, c] - v['l_r'][p][:, c]
        assert float(x0.abs().max()) == 0.0 and float(v['l_d'][p][:, c].abs().max()) == 0.0


def encoder_tolerance(xn, res, base=3e-6):
    """tests/test_gpu_parity.py:encoder_tolerance — the reference EXTRAPOLATES outside the box (offset from the clipped
    corner), trilinear weights grow like prod(1 + 2 cells_outside) and cancel; inside the box the factor is 1."""
    oob = np.maximum(np.maximum(-xn, xn - 1.0), 0.0)
    cells = oob[:, None, :] * (np.asarray(res, np.float32)[None, :, None] - 1)
    return base * np.prod(1.0 + 2.0 * cells, axis=-1) * 4.0


def test_row_sum_xcd_encoder_vs_oracle_real_tables(fr):
    """k_part_encode_rs_xcd's output for >= 1e5 (point, part) pairs of the frame against the oracle's hash_embed on
    the trainable 64-byte rows of the real (1.09 GB) tables."""
    f, k = fr, fr['k']
    v, st, net = f['v'], f['st'], f['net']
    sd = {k_: t.detach().cpu() for k_, t in net.state_dict().items()}
    model = O.Model(sd, f['cfg'])
    total, inside_total = 0, 0
    g = torch.Generator().manual_seed(7 + k)
    # (1) EVERY listed pair of the frame against the generic thread-per-point encoder on the 64-byte rows (invr_grid_encode_fwd,
    #     pinned to the reference goldens by test_grid_encoder_variants): catches rare index-arithmetic cases a sample misses
    for p in range(5):
        cnt = int(st[1 + p])
        e = net.tpose_human.part_networks[p].embedder
        x_all = v['l_x'][p][:, :cnt].t().contiguous()
        ref_all = e(x_all)                                                  # (cnt, 19)
        got_all = v['emb'][p][:19, :cnt].t()
        xn = ref_all[:, :3]
        oob = torch.clamp(torch.maximum(-xn, xn - 1.0), min=0.0)            # encoder_tolerance on the device
        res_t = torch.tensor(model.pspec[p]['res'], device=DEV, dtype=torch.float32)
        tol_all = 1.2e-5 * torch.prod(1.0 + 2.0 * oob[:, None, :] * (res_t[None, :, None] - 1), dim=-1)
        bad = ((got_all[:, 3:] - ref_all[:, 3:]).abs() > tol_all)
        if bool(bad.any()):
            i, l = [int(t[0]) for t in bad.nonzero(as_tuple=True)]
            raise AssertionError('pose %d part %d: %d mismatching (pair, level) entries of %d pairs; first: pair %d level %d x_norm %s got %r ref %r'
                                 % (k, p, int(bad.sum()), cnt, i, l, xn[i].tolist(), float(got_all[i, 3 + l]), float(ref_all[i, 3 + l])))
        assert float((got_all[:, :3] - xn).abs().max()) < 1e-6
    # (2) a sample against the oracle on the CPU
    for p in range(5):
        cnt = int(st[1 + p])                                  # incl. the far-constant pair (canonical origin, far outside most boxes)
        take = min(cnt, MIN['enc_take'])
        sel = torch.randperm(cnt, generator=g)[:take]
        sel[0] = cnt - 1
        sel = sel.to(DEV)
        x = v['l_x'][p][:, sel].t().contiguous().cpu()
        got = v['emb'][p][:19, sel].t().cpu().numpy()
        prefix = 'tpose_human.part_networks.%d.embedder.' % p
        ref = []
        with torch.no_grad():
            for c0 in range(0, take, 10000):
                ref.append(O.hash_embed(x[c0:c0 + 10000], sd, prefix, model.pspec[p]))
        ref = torch.cat(ref).numpy()
        assert np.abs(got[:, :3] - ref[:, :3]).max() < 1e-6, p
        tol = encoder_tolerance(ref[:, :3], model.pspec[p]['res'])
        err = np.abs(got[:, 3:] - ref[:, 3:])
        assert (err <= tol).all(), (k, p, float((err / tol).max()))
        inside = (tol <= 1.3e-5).all(1)
        if inside.any():
            assert err[inside].max() < 1.3e-5, (k, p, float(err[inside].max()))
        total += take
        inside_total += int(inside.sum())
    assert total >= MIN['enc_total'] and inside_total >= MIN['enc_inside']


def test_render_strict_1e4_on_well_conditioned_pixels(fr):
    """256 rays x 128 samples per pose through the production pipeline against the oracle (full tables).  Survivor /
    flag decisions identical.  Every pixel whose reference arithmetic is well conditioned (the oracle's own fp32 result
    is within 2e-6 of its fp64 result AND the fp64 result moves by less than 2e-6 under an fp32-ulp perturbation of the
    rays) must meet the plain 1e-4 bar with no allowance; the remaining pixels (far / band pairs extrapolated by the
    encoder with 1e3..1e6 weights) may deviate from the fp64 result by 1e-4 + 4x that noise scale."""
    f, k = fr, fr['k']
    net, bc, gb, cfg = f['net'], f['bc'], f['gb'], f['cfg']
    n = gb['ray_o'].shape[1]
    sel = torch.randperm(n, generator=torch.Generator().manual_seed(100 + k))[:MIN['strict_rays']].sort()[0]
    out = net.render_rays(f['ctx'], *[gb[k_][0][sel.to(DEV)] for k_ in ('ray_o', 'ray_d', 'near', 'far')], S, want_raw=True)
    torch.cuda.synchronize()
    net._ws = None
    sd = {k_: t.detach().cpu() for k_, t in net.state_dict().items()}
    b = dict(bc)
    for k_ in ('ray_o', 'ray_d', 'near', 'far'):
        b[k_] = bc[k_][:, sel]
    with torch.no_grad():
        ref = O.render(O.Model(sd, cfg), b, n_samples=S, chunk=64)
        sd64 = {k_: (t.double() if t.is_floating_point() else t) for k_, t in sd.items()}
        b64 = {k_: (t.double() if torch.is_tensor(t) and t.is_floating_point() else t) for k_, t in b.items()}
        ref64 = O.render(O.Model(sd64, cfg), b64, n_samples=S, chunk=64)
    assert int((ref['occ'][0, :, 0] != 0).sum()) > MIN['occ']
    assert bool(((out['raw'].cpu()[:, 3] != 0) == (ref['raw'][0, :, 3] != 0)).all())
    exact = ref64['rgb_map'][0]
    err_gpu = (out['rgb_map'].cpu().double() - exact).abs().max(1)[0]
    err_ref = (ref['rgb_map'][0].double() - exact).abs().max(1)[0]
    # conditioning of a pixel, independent of one particular rounding sequence (tests/conditioning.py): the largest move of the fp64
    # result under several fp32-ulp-sized random perturbations of the rays, and the oracle's own fp32 deviation
    from tests.conditioning import pixel_noise
    noise = pixel_noise(O, O.Model(sd64, cfg), b64, exact, S, ref32=ref['rgb_map'][0], seed=k,
                        rerun32=lambda ch: O.render(O.Model(sd, cfg), b, n_samples=S, chunk=ch)['rgb_map'][0])
    sens = noise
    well = noise < 2e-6
    print('pose %d: well-conditioned pixels %d / %d; max err on them %.2e; worst pixel err %.2e (noise %.2e)'
          % (k, int(well.sum()), well.numel(), float(err_gpu[well].max()) if bool(well.any()) else 0.0, float(err_gpu.max()), float(noise.max())))
    assert int(well.sum()) >= WELL_FLOOR * well.numel(), int(well.sum())
    assert float(err_gpu[well].max()) <= 1e-4, ('strict', float(err_gpu[well].max()))           # strict, no allowance
    # (4 x again, round 6: with the noise scale taken over the extra fp32 re-runs of tests/conditioning.py no pixel of the five poses
    # needed ANY allowance in five runs — max((err - 1e-4) / noise) = -1.5 .. -7.0, profiles/r6_tolerance_headroom.txt; round 5 had
    # let it out to 8 x)
    ill = ~well
    if bool(ill.any()):
        _tol_report('strict pose %d: ill-conditioned pixels %d, needed multiple of the noise scale max((err - 1e-4) / noise) = %.3f'
                    % (k, int(ill.sum()), float(((err_gpu[ill] - 1e-4) / noise[ill]).max())))
    worst = (err_gpu - (1e-4 + 4 * noise)).argmax()
    assert bool((err_gpu <= 1e-4 + 4 * noise).all()), ('ill-conditioned', float(err_gpu[worst]), float(err_ref[worst]), float(sens[worst]))
    assert float(err_gpu.median()) < 2e-6


def test_two_phase_mlp_and_winner_lists_vs_whole_field(fr):
    """The part MLPs run in two phases since round 3 (k_part_occ_all -> k_winner_lists -> k_part_rgb_all: colour only for the pair
    that wins its survivor's max-occupancy merge).  For the WHOLE frame (2.4 M survivors, 584 slot groups: the colour kernel's
    segment cursor crosses several 64-group windows) at every pose:
      * occp of every listed pair = the occupancy of the one-kernel field evaluation (invr_part_field_fwd: encoder + both MLPs for
        every pair) of the same canonical points,
      * wsel of every survivor = the reference's merge rule (inb_part_network_multiassign.py:229-256: zeros for unflagged parts,
        the part constant for far pairs, FIRST maximum) applied to those occupancies,
      * the winner lists hold exactly the winning listed pairs (+ the far-constant pair of every part), segment by segment,
      * rgbw at every survivor with a listed winner = the field's [rgb, occ] of the winning pair."""
    import ctypes as C
    f = fr
    v, st, Na, net, cfg = f['v'], f['st'], f['Na'], f['net'], f['cfg']
    L = _abi.lib()
    keep = []
    old = net.cfg
    net.cfg = cfg
    try:
        model = net.model_struct(keep)
    finally:
        net.cfg = old
    li = f['gb']['latent_index'].reshape(-1)[:1].to(torch.int64).contiguous()
    cap, lc = v['cap'], v['lcap']
    cnt = [int(st[1 + p]) for p in range(5)]
    fields = []
    cand = torch.zeros(Na, 5, device=DEV)
    kind = torch.full((Na, 5), 255, dtype=torch.int64, device=DEV)          # what wsel would say if part p won
    pair_of = torch.full((Na, 5), -1, dtype=torch.int64, device=DEV)
    fl = v['pflags'][:Na].to(torch.int64)
    ff = v['farflags'][:Na].to(torch.int64)
    worst_occ = 0.0
    for p in range(5):
        n = cnt[p]
        x = v['l_x'][p][:, :n].t().contiguous()
        d = v['l_d'][p][:, :n].t().contiguous()
        raw = torch.empty(n, 4, device=DEV)
        nb = L.invr_part_field_workspace(n)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _abi.check(L.invr_part_field_fwd(C.byref(model), p, _abi.ptr(li, torch.int64), _abi.ptr(x), _abi.ptr(d), n,
                                         _abi.ptr(raw), C.c_void_p(ws.data_ptr()), nb, _abi.stream_ptr()))
        fields.append(raw)
        occp = v['occp'][p][:n]
        worst_occ = max(worst_occ, float((raw[:, 3] - occp).abs().max()))
        slots = v['l_slot'][p][:n].long()
        assert int(slots[-1]) == cap                                         # the far-constant pair closes the list
        real = slots[:-1]
        cand[real, p] = occp[:-1]
        kind[real, p] = p
        pair_of[real, p] = torch.arange(n - 1, device=DEV)
        far = ((ff >> p) & 1).bool()
        assert not bool((far & ((fl >> p) & 1).bool()).any())
        cand[far, p] = occp[-1]
        kind[far, p] = 8 + p
    assert worst_occ <= 1e-6, worst_occ
    # the merge rule: start from part 0 whatever it is, a later part only takes over with a strictly larger occupancy
    best = cand[:, 0].clone()
    bsel = kind[:, 0].clone()
    bpair = pair_of[:, 0].clone()
    bpart = torch.zeros(Na, dtype=torch.int64, device=DEV)
    for p in range(1, 5):
        take = cand[:, p] > best
        best = torch.where(take, cand[:, p], best)
        bsel = torch.where(take, kind[:, p], bsel)
        bpair = torch.where(take, pair_of[:, p], bpair)
        bpart = torch.where(take, torch.full_like(bpart, p), bpart)
    wsel = v['wsel'][:Na].to(torch.int64)
    assert torch.equal(wsel, bsel), int((wsel != bsel).sum())
    listed = bsel < 5
    assert int(listed.sum()) > Na // 2
    # winner lists, segment by segment
    g_last = (max(Na, 1) - 1) // 4096
    wcnt = v['wcnt'][:g_last + 1].to(torch.int64)
    worst_rgb = 0.0
    for p in range(5):
        slots = v['l_slot'][p][:cnt[p]].long()
        real = slots[:-1]
        # list offset of every group = pairs of the groups before it
        per_group = torch.bincount(real // 4096, minlength=g_last + 1)
        off = torch.cumsum(per_group, 0) - per_group
        want = (listed & (bpart == p)).nonzero(as_tuple=True)[0]             # winning survivors of part p, ascending
        want_pairs = bpair[want]
        want_per_group = torch.bincount(want // 4096, minlength=g_last + 1)
        exp_cnt = want_per_group.clone()
        exp_cnt[g_last] += 1                                                 # + the far-constant pair
        assert torch.equal(wcnt[:, p], exp_cnt), p
        # gather the lists' entries: positions off[g] + [0, wcnt[g][p])
        gidx = torch.repeat_interleave(torch.arange(g_last + 1, device=DEV), wcnt[:, p])
        start = torch.cumsum(wcnt[:, p], 0) - wcnt[:, p]
        pos = off[gidx] + (torch.arange(gidx.numel(), device=DEV) - start[gidx])
        got = v['wl'][p][pos].long()
        assert int(got[-1]) == cnt[p] - 1                                    # the constant pair, last entry of the last segment
        assert torch.equal(got[:-1], want_pairs), p
        # colour of the winners
        rw = v['rgbw'][want]
        ref = fields[p][want_pairs]
        if want.numel():
            worst_rgb = max(worst_rgb, float((rw - ref).abs().max()))
        worst_rgb = max(worst_rgb, float((v['rgbw'][lc + p] - fields[p][-1]).abs().max()))     # far constant of the part
    print('pose %d: %d survivors, %d listed winners; max |occp - field occ| %.1e, max |rgbw - field raw| %.1e'
          % (f['k'], Na, int(listed.sum()), worst_occ, worst_rgb))
    assert worst_rgb <= 1e-6, worst_rgb
