"""Generate golden vectors by IMPORTING the reference (this container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; never runs on the GPU box)

The reference is pure PyTorch on this path, so it imports here on CPU after a few
stand-ins for packages this image lacks (none of which carry arithmetic of the
path, except the KNN — see below):
  * termcolor / colored_traceback / tensorboardX : logging cosmetics
  * pytorch3d.ops.knn.knn_points : brute-force squared-L2 K-NN honouring ``lengths2``
    (pytorch3d==0.7.2 is an un-vendored dependency; semantics per its docs:
    squared distances, first lengths2[n] reference points only).  No reference test
    pins this boundary -> the KNN stage is "parity unpinned" by the reference and is
    pinned by this stand-in instead.
  * torch.tensor(device='cuda') / Tensor.cuda : module-level CUDA constants
    (lib/utils/blend_utils.py:248-290, lib/networks/embedder.py:12)
Only inputs (seeds) and outputs (arrays) are stored in the .npz; no reference source.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TABLE_LOG2 = 12


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod('termcolor', colored=lambda s, *a, **k: s, cprint=lambda *a, **k: None)
    ct = mod('colored_traceback'); ct.auto = mod('colored_traceback.auto')

    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def __getattr__(self, k): return lambda *a, **kw: None
    mod('tensorboardX', SummaryWriter=SummaryWriter)

    def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, return_sorted=True, **kw):
        d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)           # (N,P1,P2) squared L2
        if lengths2 is not None:
            ar = torch.arange(p2.shape[1])[None, None, :]
            d = d.masked_fill(ar >= lengths2[:, None, None], float('inf'))
        dists, idx = d.topk(K, dim=-1, largest=False)
        r = types.SimpleNamespace(dists=dists, idx=idx, knn=None)
        return r
    p3 = mod('pytorch3d'); ops = mod('pytorch3d.ops'); knn = mod('pytorch3d.ops.knn', knn_points=knn_points)
    p3.ops = ops; ops.knn = knn; ops.knn_points = knn_points

    _tensor = torch.tensor
    def tensor(*a, **k):
        if str(k.get('device', '')).startswith('cuda'):
            k['device'] = 'cpu'
        return _tensor(*a, **k)
    torch.tensor = tensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def import_reference(n_samples, extra_opts=()):
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    opts = ['N_samples', str(n_samples), 'silent', 'True', 'use_lpips', 'False']
    for p in ['body', 'leg', 'head', 'larm', 'rarm']:
        opts += ['partnet.%s.embedder.kwargs.log2_hashmap_size' % p, str(TABLE_LOG2)]
    opts += list(extra_opts)
    sys.argv = ['make_golden', '--cfg_file', 'configs/inb/inb_377.yaml'] + opts
    from lib.config import cfg as rcfg
    return rcfg


def tnp(x):
    return x.detach().cpu().numpy()


def main():
    import invr
    from invr import scene, params
    from invr.config import make_cfg

    S = 32
    rcfg = import_reference(S)
    torch.manual_seed(0)
    from lib.networks.make_network import make_network
    from lib.networks.renderer.make_renderer import make_renderer
    from lib.utils import blend_utils as bu

    cfg = make_cfg(table_log2=TABLE_LOG2, N_samples=S)
    net = make_network(rcfg)
    sd = params.init_state_dict(cfg, seed=7)
    ref_sd = net.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), (set(ref_sd) ^ set(sd))
    for k in sd:
        assert ref_sd[k].shape == sd[k].shape and ref_sd[k].dtype == sd[k].dtype, (k, ref_sd[k].shape, sd[k].shape)
    net.load_state_dict(sd, strict=True)
    net.eval()

    # known-answer facts at FULL inb_377 sizes (no allocation: constructor arithmetic only)
    full = make_cfg()
    facts = {}
    for name in ['body', 'leg', 'head', 'larm', 'rarm']:
        sp = params.part_grid_spec(full, name)
        facts[name] = (sp['start_hash'], sp['T'], sp['dense_rows'], sp['n_hash'])
    print('full-size facts', facts)

    out = {}
    batch_np, extras = scene.make_scene(64, 64, seed=0)
    batch = scene.to_torch(batch_np)
    renderer = make_renderer(rcfg, net)

    # ---- (vi) full eval render, 64x64 rays x 32 samples (BASELINE config 1)
    with torch.no_grad():
        ret = renderer.render(dict(batch))
    out['render_rgb_map'] = tnp(ret['rgb_map'])
    out['render_acc_map'] = tnp(ret['acc_map'])
    raw = tnp(ret['raw'])[0]
    out['render_raw_nz_idx'] = np.nonzero(raw[:, 3] != 0)[0].astype(np.int32)
    out['render_raw_nz'] = raw[out['render_raw_nz_idx']]
    out['render_n_active_samples'] = np.int64((tnp(ret['occ'])[0, :, 0] != 0).sum())

    # ---- per-stage vectors on a subset of sample points of that render
    rng = np.random.RandomState(5)
    ro, rd = batch['ray_o'][0], batch['ray_d'][0]
    nr = ro.shape[0]
    sel = np.sort(rng.choice(nr, 96, replace=False))
    out['sel_rays'] = sel.astype(np.int32)
    wpts, z = renderer.get_wsampling_points(batch['ray_o'][:, sel], batch['ray_d'][:, sel],
                                            batch['near'][:, sel], batch['far'][:, sel])
    out['wpts'] = tnp(wpts)
    out['z_vals'] = tnp(z)
    wflat = wpts.reshape(1, -1, 3)
    pose_pts = bu.world_points_to_pose_points(wflat, batch['R'], batch['Th'])
    pose_dirs = bu.world_dirs_to_pose_dirs(batch['ray_d'][:, sel][:, :, None].expand(-1, -1, S, -1).reshape(1, -1, 3), batch['R'])
    out['pose_pts'] = tnp(pose_pts)
    # (ii) volume sampling
    pnorm = bu.pts_sample_blend_weights(pose_pts, batch['pbw'][..., -1:], batch['pbounds'])[0, -1]
    out['pnorm'] = tnp(pnorm)
    act = (pnorm < rcfg.smpl_thresh).nonzero(as_tuple=True)[0]
    out['active_idx'] = tnp(act).astype(np.int32)
    ap = pose_pts[:, act]
    ad = pose_dirs[:, act]
    # (iii) KNN blend
    with torch.no_grad():
        knn_bw = bu.pts_knn_blend_weights_multiassign_batch(ap, batch['part_pts'][0], batch['part_pbw'][0], batch['lengths2'][0])
    out['knn_bw'] = tnp(knn_bw)
    # (iv) warp + deformer
    with torch.no_grad():
        tpose, tdirs, resd, pflag, init_big, pnorm2 = net.pose_points_to_tpose_points(ap, ad, batch)
    out['tpose'] = tnp(tpose); out['tpose_dirs'] = tnp(tdirs); out['resd'] = tnp(resd)
    out['pflag'] = tnp(pflag); out['init_bigpose'] = tnp(init_big)
    tb = batch['tbounds'][0]
    upts = torch.rand(1, 500, 3, generator=torch.Generator().manual_seed(1)) * (tb[1] - tb[0]) * 1.2 + tb[0] - 0.1 * (tb[1] - tb[0])
    out['uv_pts'] = tnp(upts)
    out['uv_out'] = tnp(bu.pts_sample_uv(upts, batch['tuv'], batch['tbounds'], mode='bilinear'))
    # (v) per-part field on the warped points of that part (+ merged output)
    with torch.no_grad():
        for pid, pn in enumerate(net.tpose_human.part_networks):
            f = pflag[0, :, pid]
            r = pn(tpose[0, f, pid], tdirs[0, f, pid], None, batch)
            out['part%d_raw' % pid] = tnp(r['raw'])
        merged = net.tpose_human(tpose[0], tdirs[0], pflag[0], None, None, batch)
    out['merged_raw'] = tnp(merged['raw']); out['merged_occ'] = tnp(merged['occ']); out['tocc'] = tnp(merged['tocc'])

    # ---- (i) embedder in/out incl. out-of-bbox points: part-style (dense+hash), deformer-style,
    #      and a start_hash==0 non-separate table (base_resolution 32 > 4099^(1/3))
    from lib.networks.embedders.part_base_embedder import Embedder
    g = torch.Generator().manual_seed(3)
    for tag, pn in [('body', net.tpose_human.part_networks[0]), ('head', net.tpose_human.part_networks[2])]:
        bb = pn.embedder.bounds
        x = torch.rand(400, 3, generator=g) * (bb[1] - bb[0]) * 1.3 + bb[0] - 0.15 * (bb[1] - bb[0])
        out['emb_%s_x' % tag] = tnp(x)
        with torch.no_grad():
            out['emb_%s_y' % tag] = tnp(pn.embedder(x, {}))
    x = torch.rand(400, 3, generator=g) * 1.3 - 0.15
    out['emb_deform_x'] = tnp(x)
    with torch.no_grad():
        out['emb_deform_y'] = tnp(net.tpose_deformer.embedder(x, {}))
    kw = dict(n_levels=6, n_features_per_level=4, log2_hashmap_size=8, base_resolution=8, b=1.38,
              sum=True, sum_over_features=False, separate_dense=True, use_batch_bounds=False)
    e0 = Embedder(bbox=np.array([[-1, -1, -1], [1, 2, 1]]), **kw)
    sp0 = params.grid_spec(bbox=[[-1, -1, -1], [1, 2, 1]], **kw)
    assert sp0['start_hash'] == 0 and not sp0['separate_dense'] and int(e0.start_hash) == 0
    r0 = np.random.RandomState(11)
    tab = (r0.standard_normal(tuple(e0.hash.shape)) * 0.1).astype(np.float32)
    e0.hash.data = torch.from_numpy(tab)
    x = torch.rand(300, 3, generator=g) * 3.6 - 1.3
    out['emb_allhash_x'] = tnp(x)
    with torch.no_grad():
        out['emb_allhash_y'] = tnp(e0(x, {}))

    # ---- (vii) train-mode forward + gradients, 256 rays, fixed jitter
    net.train()
    tsel = np.sort(np.random.RandomState(9).choice(nr, 256, replace=False))
    out['train_rays'] = tsel.astype(np.int32)
    tb_ = {k: v for k, v in batch.items()}
    for k in ['ray_o', 'ray_d', 'near', 'far', 'rgb', 'occupancy']:
        tb_[k] = batch[k][:, tsel]
    jit = torch.rand(1, 256, S, generator=torch.Generator().manual_seed(21))
    pair = {}
    _rand, _rand_like = torch.rand, torch.rand_like
    def rand(*a, **k):
        shape = a[0] if len(a) == 1 and not isinstance(a[0], int) else a
        assert tuple(shape) == tuple(jit.shape), shape
        return jit.clone()
    def rand_like(x, **k):
        pair['u'] = _rand(x.shape, generator=torch.Generator().manual_seed(22))
        return pair['u'].clone()
    torch.rand, torch.rand_like = rand, rand_like
    try:
        tret = renderer.render(dict(tb_))
    finally:
        torch.rand, torch.rand_like = _rand, _rand_like
    out['train_jitter'] = tnp(jit)
    out['train_pair_u'] = tnp(pair['u']) if 'u' in pair else np.zeros((1, 0, 3), np.float32)
    for k in ['rgb_map', 'acc_map', 'resd', 'tpts', 'tocc', 'oresd', 'reg_distortion_loss']:
        out['train_' + k] = tnp(tret[k])
    loss = ((tret['rgb_map'] - tb_['rgb']) ** 2).mean() + 0.1 * tret['reg_distortion_loss'].mean() \
        + 0.1 * torch.norm(tret['resd'], dim=2).mean()
    loss.backward()
    out['train_loss'] = tnp(loss)
    gsd = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    for k, gten in gsd.items():
        gnp = tnp(gten)
        if gnp.size > 4096:     # big tables: store the touched rows sparsely (row norm > 0)
            flat = gnp.reshape(-1, gnp.shape[-1])
            rows = np.nonzero(np.abs(flat).sum(1) > 0)[0]
            out['grad_rows::' + k] = rows.astype(np.int32)
            out['grad_vals::' + k] = flat[rows][:, :1].copy()     # all F columns are equal (sum over features)
            out['grad_full_equal::' + k] = np.array(bool(np.all(flat[rows] == flat[rows][:, :1]))) if flat.shape[1] > 2 else np.array(False)
            if not out['grad_full_equal::' + k]:
                out['grad_vals::' + k] = flat[rows]
        else:
            out['grad::' + k] = gnp
    meta = dict(table_log2=TABLE_LOG2, n_samples=S, scene_seed=0, param_seed=7, H=64, W=64,
                smpl_thresh=float(rcfg.smpl_thresh))
    out['meta_keys'] = np.array(list(meta.keys()))
    out['meta_vals'] = np.array([float(v) for v in meta.values()])
    out['facts_names'] = np.array(list(facts.keys()))
    out['facts'] = np.array([facts[k] for k in facts], dtype=np.int64)
    out['ref_n_params_reduced'] = np.int64(sum(p.numel() for p in net.parameters()))
    path = os.path.join(HERE, 'inb377_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) / 1e6, 'MB;', len(out), 'arrays')


if __name__ == '__main__':
    main()
