"""Pin the oracle (oracle/nvr_oracle.py) against vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import nvr_oracle as O
from invr import params
from invr.config import make_cfg, PART_NAMES

TOL = 2e-6


def close(a, b, tol=TOL):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol, err


def test_known_answer_table_geometry(golden):
    """start_hash / prime / row counts at full inb_377 sizes (SURVEY.md §8c known-answer facts)."""
    full = make_cfg()
    for name, row in zip(golden['facts_names'].tolist(), golden['facts']):
        sp = params.part_grid_spec(full, name)
        assert (sp['start_hash'], sp['T'], sp['dense_rows'], sp['n_hash']) == tuple(int(v) for v in row)
    expect = {'body': (6, 1048583), 'leg': (13, 1048583), 'head': (11, 262147), 'larm': (9, 32771), 'rarm': (9, 32771)}
    for n, (sh, T) in expect.items():
        sp = params.part_grid_spec(full, n)
        assert sp['start_hash'] == sh and sp['T'] == T
    d = params.deformer_grid_spec(full)
    assert d['start_hash'] == 6 and d['T'] == 16411 and d['dense_rows'] == 12276
    # total parameter count of the reference network at inb_377 defaults
    tot = 0
    for n in PART_NAMES:
        sp = params.part_grid_spec(full, n)
        occ, rgb = params.mlp_dims(full, n)
        tot += sp['dense_rows'] * 16 + sp['n_hash'] * sp['T'] * 16 + 6 + 4 * 16 + 16 + 24 + 8 + 100 * 8
        tot += sum(a * b + b for a, b in zip(occ[:-1], occ[1:])) + sum(a * b + b for a, b in zip(rgb[:-1], rgb[1:]))
    tot += d['dense_rows'] * 2 + d['n_hash'] * d['T'] * 2 + 6 + 4 * 8 + 8 + 24 + (19 * 32 + 32 + 32 * 32 + 32 + 32 * 3 + 3)
    assert tot == 285993711


def test_oracle_geometry_is_its_own_restatement(golden):
    """The oracle's embedder_geometry (its own restatement of part_base_embedder.py:13-104) against (i) the facts the golden
    generator read off the IMPORTED reference's embedders at full inb_377 size, (ii) sympy.nextprime — the function the reference
    calls —, (iii) the product's separate restatement (invr.params.grid_spec), field by field, (iv) the reference's state_dict
    tensors entries_size / entries_num / entries_cnt / entries_sum of the small golden model.  The oracle imports nothing of
    the product."""
    import inspect
    src = inspect.getsource(O)
    assert 'import invr' not in src and 'from invr' not in src
    full = make_cfg()
    for name, row in zip(golden['facts_names'].tolist(), golden['facts']):
        g = O.embedder_geometry(bbox=full.partnet[name].bbox, **full.partnet[name].embedder.kwargs)
        assert (g['start_hash'], g['T'], g['dense_rows'], g['n_hash']) == tuple(int(v) for v in row)
    try:
        from sympy import nextprime
        for e in (4, 10, 12, 14, 15, 18, 19, 20, 22):
            assert O.next_prime(2 ** e) == int(nextprime(2 ** e)) == params.next_prime(2 ** e)
    except ImportError:
        assert [O.next_prime(2 ** e) for e in (14, 15, 18, 20)] == [16411, 32771, 262147, 1048583]      # SURVEY 8(c), probed
    for cfg in (full, make_cfg(table_log2=12)):
        specs = [(O.embedder_geometry(**cfg.tpose_deformer.embedder.kwargs), params.deformer_grid_spec(cfg))]
        specs += [(O.embedder_geometry(bbox=cfg.partnet[n].bbox, **cfg.partnet[n].embedder.kwargs), params.part_grid_spec(cfg, n)) for n in PART_NAMES]
        for g, sp in specs:
            for k in ('L', 'F', 'T', 'res', 'cnt', 'start_hash', 'separate_dense', 'dense_rows', 'n_hash', 'sum', 'sum_over_features',
                      'include_input', 'out_dim'):
                assert g[k] == sp[k], k
            assert np.array_equal(g['size'].numpy(), sp['size']) and np.array_equal(g['bbox'].numpy(), sp['bbox'])
    # the start_hash == 0 branch (:69: separate_dense forced off)
    g0 = O.embedder_geometry(base_resolution=20, log2_hashmap_size=12)
    assert g0['start_hash'] == 0 and not g0['separate_dense'] and g0['n_hash'] == 16


def test_oracle_geometry_vs_reference_state_dict(golden, small_setup):
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    for i, g in enumerate(m.pspec):
        pre = 'tpose_human.part_networks.%d.embedder.' % i
        assert torch.equal(sd[pre + 'entries_num'].long(), torch.tensor(g['res']))
        assert torch.equal(sd[pre + 'entries_cnt'].long(), torch.tensor(g['cnt']))
        assert torch.equal(sd[pre + 'entries_sum'].long(), g['entries_sum'])
        assert torch.equal(sd[pre + 'entries_size'], g['size'])
        assert tuple(sd[pre + 'hash'].shape) == (g['n_hash'], g['T'], g['F'])
    assert m.n_occ == [2] * 5 and m.n_rgb == [3, 2, 3, 2, 2]


def test_sampling_and_volumes(golden, small_setup):
    cfg, sd, batch, _ = small_setup
    sel = torch.from_numpy(golden['sel_rays'].astype(np.int64))
    pts, z = O.sample_points(batch['ray_o'][:, sel], batch['ray_d'][:, sel], batch['near'][:, sel],
                             batch['far'][:, sel], cfg.N_samples)
    close(pts, golden['wpts']); close(z, golden['z_vals'])
    pp = O.world_to_pose(pts.reshape(-1, 3), batch['R'][0], batch['Th'][0])
    close(pp[None], golden['pose_pts'])
    pn = O.sample_volume(pp, batch['pbw'][0][..., -1:], batch['pbounds'][0])[:, 0]
    close(pn, golden['pnorm'])
    act = (pn < cfg.smpl_thresh).nonzero(as_tuple=True)[0]
    assert np.array_equal(act.numpy(), golden['active_idx'])
    uv = O.sample_volume(torch.from_numpy(golden['uv_pts'][0]), batch['tuv'][0], batch['tbounds'][0])
    close(uv.t()[None], golden['uv_out'])


def test_knn_warp_deformer(golden, small_setup):
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    b = {k: v[0] for k, v in batch.items()}
    pp = torch.from_numpy(golden['pose_pts'][0])
    act = torch.from_numpy(golden['active_idx'].astype(np.int64))
    ap = pp[act]
    S = cfg.N_samples
    sel = torch.from_numpy(golden['sel_rays'].astype(np.int64))
    pd = O.world_dirs_to_pose(b['ray_d'][sel][:, None].expand(-1, S, -1).reshape(-1, 3), b['R'])[act]
    bw, dist = O.knn_blend(ap, b['part_pts'], b['part_pbw'], b['lengths2'])
    close(torch.cat([bw, dist[..., None]], -1)[None], golden['knn_bw'])
    pflag = dist < cfg.smpl_thresh
    assert np.array_equal(pflag.numpy()[None], golden['pflag'])
    x_b, d_b = O.lbs_warp(ap, pd, bw, b['A'], b['big_A'])
    close(x_b.reshape(1, -1, 3), golden['init_bigpose'], 5e-6)
    close(d_b[None], golden['tpose_dirs'], 5e-6)
    resd = torch.zeros_like(x_b)
    resd[pflag] = O.deformer(x_b[pflag], sd, m.dspec, b['tuv'], b['tbounds'], b['frame_dim'])
    close(resd[None], golden['resd'])
    close((x_b + resd)[None], golden['tpose'], 5e-6)


def test_hash_embedder_variants(golden, small_setup):
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    for tag, pid in (('body', 0), ('head', 2)):
        y = O.hash_embed(torch.from_numpy(golden['emb_%s_x' % tag]), sd,
                         'tpose_human.part_networks.%d.embedder.' % pid, m.pspec[pid])
        close(y, golden['emb_%s_y' % tag], 5e-6)
    y = O.hash_embed(torch.from_numpy(golden['emb_deform_x']), sd, 'tpose_deformer.embedder.', m.dspec)
    close(y, golden['emb_deform_y'])
    # start_hash == 0 -> single (L,T,F) table, sum over levels (not features)
    kw = dict(n_levels=6, n_features_per_level=4, log2_hashmap_size=8, base_resolution=8, b=1.38,
              sum=True, sum_over_features=False, separate_dense=True, use_batch_bounds=False)
    sp = params.grid_spec(bbox=[[-1, -1, -1], [1, 2, 1]], **kw)
    assert sp['start_hash'] == 0 and not sp['separate_dense']
    tab = (np.random.RandomState(11).standard_normal((sp['L'], sp['T'], sp['F'])) * 0.1).astype(np.float32)
    sd0 = {'e.bounds': torch.from_numpy(sp['bbox']), 'e.entries_size': torch.from_numpy(sp['size']),
           'e.entries_num': torch.tensor(sp['res']), 'e.offsets': torch.from_numpy(params.CORNER_OFFSETS),
           'e.hash': torch.from_numpy(tab)}
    close(O.hash_embed(torch.from_numpy(golden['emb_allhash_x']), sd0, 'e.', sp), golden['emb_allhash_y'], 5e-6)


def test_part_fields_and_merge(golden, small_setup):
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    tpose = torch.from_numpy(golden['tpose'][0]); tdirs = torch.from_numpy(golden['tpose_dirs'][0])
    pflag = torch.from_numpy(golden['pflag'][0])
    raws = torch.zeros(tpose.shape[0], 5, 4)
    for pid in range(5):
        f = pflag[:, pid]
        r = O.part_field(tpose[f, pid], tdirs[f, pid], sd, pid, m.pspec[pid], m.n_occ[pid], m.n_rgb[pid],
                         batch['latent_index'][0], m.n_freq)
        close(r, golden['part%d_raw' % pid], 5e-6)
        raws[f, pid] = r
    raw, occ = O.merge_parts(raws)
    close(raw, golden['merged_raw'], 5e-6); close(occ[:, None], golden['merged_occ'], 5e-6)
    close(raws[..., 3:], golden['tocc'], 5e-6)


def test_full_render_64x64x32(golden, small_setup):
    """BASELINE config 1: 64x64 image, 32 samples per ray, eval mode."""
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    with torch.no_grad():
        r = O.render(m, batch)
    close(r['rgb_map'], golden['render_rgb_map'], 5e-6)
    close(r['acc_map'], golden['render_acc_map'], 5e-6)
    raw = r['raw'][0].numpy()
    nz = np.nonzero(raw[:, 3] != 0)[0]
    assert np.array_equal(nz, golden['render_raw_nz_idx'])
    close(raw[nz], golden['render_raw_nz'], 5e-6)
    # chunked == unchunked (inb_renderer.py:217-237)
    with torch.no_grad():
        rc = O.render(m, batch, chunk=512)
    close(rc['rgb_map'], r['rgb_map'].numpy(), 1e-6)


def test_train_mode_forward(golden, small_setup):
    """Train-mode quantities (stratified jitter, distortion loss, resd/tocc layouts)."""
    cfg, sd, batch, _ = small_setup
    m = O.Model(sd, cfg)
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64))
    tb = dict(batch)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        tb[k] = batch[k][:, tsel]
    with torch.no_grad():
        r = O.render(m, tb, jitter=torch.from_numpy(golden['train_jitter']), want_train=True)
    close(r['rgb_map'], golden['train_rgb_map'], 5e-6)
    close(r['resd'].reshape(1, -1, 3), golden['train_resd'])
    close(r['tocc'].reshape(1, -1, 1), golden['train_tocc'], 5e-6)
    close(O.distortion_loss(r['weights'], r['z'])[None], golden['train_reg_distortion_loss'], 5e-6)


MODE_CFG = {'mean': dict(aggr='mean'), 'rbg': dict(random_bg=True), 'dist': dict(aggr='dist'), 'mind': dict(aggr='mindist')}


def mode_train_loss(r, rgb):
    """the loss make_golden_modes.py differentiates: mse + 0.1 mean(distortion) + 0.1 mean |resd|"""
    return ((r['rgb_map'] - rgb) ** 2).mean() + 0.1 * O.distortion_loss(r['weights'], r['z']).mean() \
        + 0.1 * torch.norm(r['resd'].reshape(1, -1, 3), dim=2).mean()


def test_non_default_modes_vs_reference(golden, golden_modes, small_setup):
    """cfg.aggr = 'mean' (inb_part_network_multiassign.py:236-239) and cfg.random_bg = True (= render_weights' epsilon,
    inb_renderer.py:72) against the imported reference (tests/golden/make_golden_modes.py): eval render, train-mode forward, loss and
    every parameter gradient; round 5: aggr = 'dist' / 'mindist' (:240-251, tests/golden/modes_dist_small.npz) the same way."""
    cfg0, sd, batch, _ = small_setup
    import copy
    tsel = torch.from_numpy(golden['train_rays'].astype(np.int64))
    for tag, over in MODE_CFG.items():
        cfg = copy.deepcopy(cfg0)
        cfg.update(over)
        scale = max(1.0, float(np.abs(golden_modes[tag + '_rgb_map']).max()))
        with torch.no_grad():
            r = O.render(O.Model(sd, cfg), batch)
        close(r['rgb_map'], golden_modes[tag + '_rgb_map'], 5e-6 * scale)
        close(r['acc_map'], golden_modes[tag + '_acc_map'], 5e-6 * scale)
        raw = r['raw'][0].numpy()
        nz = golden_modes[tag + '_raw_nz_idx']
        assert np.abs(raw[nz] - golden_modes[tag + '_raw_nz']).max() < 5e-6
        mask = np.ones(raw.shape[0], bool)
        mask[nz] = False
        assert np.abs(raw[mask]).max() == 0.0
        assert np.abs(golden_modes[tag + '_rgb_map'] - golden['render_rgb_map']).max() > 1e-3       # the switch does change the image
        # train mode: forward, loss, gradients
        leaves = {k: (v.clone().requires_grad_() if v.is_floating_point() and ('entries' not in k and 'offsets' not in k and 'bounds' not in k) else v)
                  for k, v in sd.items()}
        tb = dict(batch)
        for k in ('ray_o', 'ray_d', 'near', 'far', 'rgb'):
            tb[k] = batch[k][:, tsel]
        rt = O.render(O.Model(leaves, cfg), tb, jitter=torch.from_numpy(golden['train_jitter']), want_train=True)
        close(rt['rgb_map'], golden_modes[tag + '_train_rgb_map'], 5e-6 * scale)
        close(rt['tocc'].reshape(1, -1, 1), golden_modes[tag + '_train_tocc'], 5e-6)
        loss = mode_train_loss(rt, tb['rgb'])
        assert abs(float(loss) - float(golden_modes[tag + '_train_loss'])) < 1e-6 * max(1.0, abs(float(loss)))
        loss.backward()
        n = 0
        for k, v in leaves.items():
            if not (torch.is_tensor(v) and v.requires_grad) or v.grad is None:
                continue
            if (tag + '_grad::' + k) in golden_modes:
                g = golden_modes[tag + '_grad::' + k]
                assert np.abs(v.grad.numpy() - g).max() <= 2e-4 * max(np.abs(g).max(), 1e-12) + 1e-9, (tag, k)
                n += 1
            elif (tag + '_grad_rows::' + k) in golden_modes:
                rows = golden_modes[tag + '_grad_rows::' + k]
                flat = v.grad.numpy().reshape(-1, v.shape[-1])
                g = golden_modes[tag + '_grad_vals::' + k]
                assert np.abs(flat[rows][:, :1] - g).max() <= 2e-4 * max(np.abs(g).max(), 1e-12) + 1e-9, (tag, k)
                n += 1
        assert n >= 60, n


def parts_inputs(seed):
    """Inputs of tests/golden/make_golden_parts.py, regenerated from the seed."""
    from invr import scene
    tverts, weights, parts, _ = scene.make_body(seed)
    rng = np.random.RandomState(seed + 5)
    poses = rng.uniform(-1, 1, (24, 3)) * 0.4
    poses[0] = 0
    A = scene.rigid_transformation(poses, scene._J, scene.PARENTS)
    big = np.zeros(72); big[5] = np.deg2rad(30); big[8] = np.deg2rad(-30)
    ppts = scene.lbs(tverts, weights, A)
    tpose = scene.lbs(tverts, weights, scene.rigid_transformation(big.reshape(24, 3), scene._J, scene.PARENTS))
    return ppts, weights, parts.astype(np.int64), tpose


def digest(v):
    v = np.asarray(v, np.float64)
    return np.array([v.sum(), np.abs(v).sum(), (v * (np.arange(v.size).reshape(v.shape) % 97)).sum()])


def check_parts_against_golden(tag, part_pts, part_pbw, lengths2, bounds, g):
    """Per-part packing results against the reference's own lines (tpose_dataset.py:569-591; parts_small.npz stores digests of
    the large arrays + 50 sampled rows + the small arrays in full)."""
    assert np.array_equal(np.asarray(lengths2), g[tag + '_lengths2'])
    assert np.array_equal(np.asarray(bounds), g[tag + '_bounds'])
    for name, arr in (('part_pts', part_pts), ('part_pbw', part_pbw)):
        arr = np.asarray(arr)
        assert tuple(arr.shape) == tuple(g[tag + '_' + name + '_shape'])
        assert np.array_equal(digest(arr), g[tag + '_' + name]), name
        flat = arr.reshape(-1, arr.shape[-1])
        assert np.array_equal(flat[:: max(1, flat.shape[0] // 50)][:50], g[tag + '_' + name + '_rows'])


def test_part_packing_restatement_vs_reference_lines():
    """invr.scene's per-part KNN reference sets (what every synthetic batch is built with) against the reference's inline code."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'parts_small.npz'))
    for tag, seed in (('a', 0), ('b', 7)):
        ppts, weights, parts, tpose = parts_inputs(seed)
        overlap = float(g[tag + '_overlap'])
        P, N = 5, ppts.shape[0]
        part_pts = np.zeros((P, N, 3), np.float32); part_pbw = np.zeros((P, N, 24), np.float32)
        lengths2 = np.zeros(P, np.int64); bounds = np.zeros((P, 2, 3), np.float32)
        for pid in range(P):
            f = parts == pid
            lengths2[pid] = f.sum()
            part_pts[pid, :lengths2[pid]] = ppts[f]
            part_pbw[pid, :lengths2[pid]] = weights[f]
            bounds[pid, 0] = tpose[f].min(0) - np.float32(overlap)
            bounds[pid, 1] = tpose[f].max(0) + np.float32(overlap)
        M = int(lengths2.max())
        check_parts_against_golden(tag, part_pts[:, :M], part_pbw[:, :M], lengths2, bounds, g)
