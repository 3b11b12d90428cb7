"""Differentiable training forward.

Two deliveries of the same gradients (tests/test_gpu_parity.py compares them with each other and with the reference's
autograd goldens):

  TrainRenderFn (default, cfg.train_fused)   ONE autograd node for the whole training forward: invr_train_fwd (geometry,
      64-byte-row encoder, part MLPs, merge + compositing, distortion, offset / pair regulariser terms reduced on the device)
      and invr_train_bwd (every transpose of it as HIP kernels, csrc/k_train.hip; DESIGN.md §6 "Training path").  With a
      GradArena (FusedAdam.attach) the backward accumulates into persistent buffers — the part grids as row-scalar gradients —
      and returns no tensors to autograd; without one it fills autograd-allocated dense gradients.
  render_train (cfg.train_fused = False)     the op-by-op graph: geometry from the HIP pipeline (no gradient in the reference
      either: ``torch.no_grad`` blocks, inb_part_network_multiassign.py:87-90, 132-140), the differentiable remainder
      recomputed on the pair lists from GridEncodeFn / PartMlpFn / CompositeFn (HIP forward + backward each) and torch glue.
      Kept as the in-repo cross-check and for callers that wrap the network in DistributedDataParallel.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _abi
from .config import NUM_PARTS


class GridEncodeFn(torch.autograd.Function):
    """HashEmbedder.forward under autograd (part_base_embedder.py:106-174)."""

    @staticmethod
    def forward(ctx, xyz, dense, hsh, bounds, spec):
        keep = []
        g = _abi.make_grid(spec, dense, hsh, bounds, keep)
        x = xyz.detach().to(torch.float32).contiguous()
        out = torch.empty(x.shape[0], spec['out_dim'], device=x.device, dtype=torch.float32)
        _abi.check(_abi.lib().invr_grid_encode_fwd(C.byref(g), _abi.ptr(x), x.shape[0], _abi.ptr(out), _abi.stream_ptr()))
        ctx.save_for_backward(x, dense, hsh, bounds)
        ctx.spec = spec
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, dense, hsh, bounds = ctx.saved_tensors
        spec = ctx.spec
        keep = []
        g = _abi.make_grid(spec, dense, hsh, bounds, keep)
        g_out = g_out.to(torch.float32).contiguous()
        g_hash = torch.zeros_like(hsh)
        g_dense = torch.zeros_like(dense) if dense is not None else None
        g_xyz = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _abi.check(_abi.lib().invr_grid_encode_bwd(C.byref(g), _abi.ptr(x), _abi.ptr(g_out), x.shape[0], _abi.ptr(g_dense),
                                                   _abi.ptr(g_hash), _abi.ptr(g_xyz), _abi.stream_ptr()))
        return g_xyz, g_dense, g_hash, None, None


class CompositeFn(torch.autograd.Function):
    """volume_rendering with epsilon 0 (net_utils.py:12-44): raw (R,S,4) -> weights, rgb_map, acc_map."""

    @staticmethod
    def forward(ctx, raw):
        raw = raw.detach().to(torch.float32).contiguous()
        R, S = raw.shape[:2]
        w = torch.empty(R, S, device=raw.device)
        rgb = torch.empty(R, 3, device=raw.device)
        acc = torch.empty(R, device=raw.device)
        _abi.check(_abi.lib().invr_composite_fwd(_abi.ptr(raw), R, S, _abi.ptr(w), _abi.ptr(rgb), _abi.ptr(acc), _abi.stream_ptr()))
        ctx.save_for_backward(raw)
        return w, rgb, acc

    @staticmethod
    def backward(ctx, g_w, g_rgb, g_acc):
        (raw,) = ctx.saved_tensors
        R, S = raw.shape[:2]
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        g_rgb = c(g_rgb) if g_rgb is not None else torch.zeros(R, 3, device=raw.device)
        g_raw = torch.empty_like(raw)
        _abi.check(_abi.lib().invr_composite_bwd(_abi.ptr(raw), _abi.ptr(g_rgb), _abi.ptr(c(g_acc)), _abi.ptr(c(g_w)), R, S,
                                                 _abi.ptr(g_raw), _abi.stream_ptr()))
        return g_raw


def sample_volume(vol, bounds, pts, c0, nc):
    """pts_sample_uv / pts_sample_blend_weights (blend_utils.py:501-555), no gradient."""
    vol = vol.contiguous()
    dims = (C.c_int32 * 3)(*vol.shape[:3])
    pts = pts.detach().to(torch.float32).contiguous()
    out = torch.empty(pts.shape[0], nc, device=pts.device)
    bounds = bounds.contiguous()                       # held in a name: the pointer must outlive the call
    _abi.check(_abi.lib().invr_sample_volume(_abi.ptr(vol), dims, vol.shape[3], c0, nc, _abi.ptr(bounds),
                                             _abi.ptr(pts), pts.shape[0], _abi.ptr(out), _abi.stream_ptr()))
    return out


_SPLIT_ROWS = 2048


class LinearFn(torch.autograd.Function):
    """nn.Linear whose weight gradient dW = dY^T X (K = tens of thousands of pairs, M x N = 64 x 70 at most) is a
    batched GEMM over 2048-row slabs + a sum: rocBLAS runs the plain tall-skinny GEMM autograd would issue on a
    handful of workgroups (140-250 us each, 5 ms per training step for the 17 linears)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        n = x.shape[0]
        main = (n // _SPLIT_ROWS) * _SPLIT_ROWS
        gw = None
        if main:
            gw = torch.bmm(gy[:main].view(-1, _SPLIT_ROWS, gy.shape[1]).transpose(1, 2), x[:main].view(-1, _SPLIT_ROWS, x.shape[1])).sum(0)
        if main < n:
            tail = gy[main:].t() @ x[main:]
            gw = tail if gw is None else gw + tail
        gx = gy @ w if ctx.needs_input_grad[0] else None
        return gx, gw, gy.sum(0)


def linear(l, x):
    return LinearFn.apply(x, l.weight, l.bias)


def mlp_forward(mlp, x):
    """part_base_network.MLP.forward (:18-24)."""
    for l in mlp.linears[:-1]:
        x = F.softplus(linear(l, x))
    return linear(mlp.linears[-1], x)


def dir_encode(d, n_freq):
    """freq_embedder.PosEnc.forward (:20-31)."""
    out = [d]
    for k in range(n_freq):
        out += [torch.sin(d * (2.0 ** k)), torch.cos(d * (2.0 ** k))]
    return torch.cat(out, -1)


def deform(net, batch, pts):
    """Deformer.forward (uv_deformer.py:31-38) with gradients to the deformer's tables and MLP."""
    dfm = net.tpose_deformer
    uv = sample_volume(batch['tuv'][0], batch['tbounds'][0], pts, 0, 2)
    t = batch['frame_dim'].reshape(1, 1).expand(uv.shape[0], 1).float()
    uvt = torch.cat([uv, t], -1)
    e = dfm.embedder
    feat = GridEncodeFn.apply(uvt, e.dense if e.separate_dense else None, e.hash, e.bounds, e.spec)
    h = F.softplus(linear(dfm.mlp[0], feat))
    h = F.softplus(linear(dfm.mlp[2], h))
    return 0.05 * torch.tanh(linear(dfm.mlp[4], h))


def _splitk(gy, x):
    """gy (n,out), x (n,in) -> gy^T x (out,in) as a slab-batched GEMM (see LinearFn)."""
    n = x.shape[0]
    main = (n // _SPLIT_ROWS) * _SPLIT_ROWS
    gw = None
    if main:
        gw = torch.bmm(gy[:main].view(-1, _SPLIT_ROWS, gy.shape[1]).transpose(1, 2), x[:main].view(-1, _SPLIT_ROWS, x.shape[1])).sum(0)
    if main < n:
        tail = gy[main:].t() @ x[main:]
        gw = tail if gw is None else gw + tail
    return gw


def _rgb1_col(s, g):
    """csrc/mlp_common.h rgb1_col: input column of rgb layer 1 held by k-slot (s, g); -1 = padding."""
    if s < 5:
        e = 4 * s + g
        return e if e < 19 else -1
    if s < 11:
        u = s - 5
        return 19 + 3 + g * 6 + (u & 1) * 3 + (u >> 1)
    if s < 14:
        e = 4 * (s - 11) + g
        return 19 + e if e < 3 else (62 + (e - 3) if e < 11 else -1)
    return 46 + 4 * g + (s - 14)


_SLOT_OF_COL = [0] * 70
for _s in range(18):
    for _g in range(4):
        if _rgb1_col(_s, _g) >= 0:
            _SLOT_OF_COL[_rgb1_col(_s, _g)] = 4 * _s + _g


_slot_cache = {}


def _slot_of_col(dev):
    if dev not in _slot_cache:
        _slot_cache[dev] = torch.as_tensor(_SLOT_OF_COL, device=dev)
    return _slot_cache[dev]


class PartMlpFn(torch.autograd.Function):
    """The two MLPs of one part (part_base_network.py:44-63 after the encoder) on the matrix cores in both directions:
    invr_part_mlp_fwd / invr_part_mlp_bwd.  The backward kernel recomputes the forward, returns the embedding gradient
    and the per-layer (g_z, a_in) matrices; the K = n weight-gradient reductions are slab-batched GEMMs here."""

    @staticmethod
    def forward(ctx, emb, dirs, model, pid, latent_index, rgb_latent, *params):
        n = emb.shape[0]
        dev = emb.device
        emb_soa = torch.zeros(20, n, device=dev)
        emb_soa[:19] = emb.detach().t()
        dirs_soa = dirs.detach().t().contiguous()
        raw = torch.empty(n, 4, device=dev)
        cnt = torch.full((1,), n, dtype=torch.int32, device=dev)
        li = latent_index.reshape(-1)[:1].to(torch.int64).contiguous()
        _abi.check(_abi.lib().invr_part_mlp_fwd(C.byref(model), pid, _abi.ptr(li, torch.int64), _abi.ptr(emb_soa), _abi.ptr(dirs_soa), n,
                                                _abi.ptr(cnt, torch.int32), _abi.ptr(raw), _abi.stream_ptr()))
        ctx.save_for_backward(emb_soa, dirs_soa, li, rgb_latent)
        ctx.model, ctx.pid, ctx.n_rgb = model, pid, (len(params) - 4) // 2
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        emb_soa, dirs_soa, li, rgb_latent = ctx.saved_tensors
        n, dev, three = emb_soa.shape[1], emb_soa.device, ctx.n_rgb == 3
        n_pad = (n + _SPLIT_ROWS - 1) // _SPLIT_ROWS * _SPLIT_ROWS
        gz = torch.zeros(5, n_pad, 64, device=dev)
        a = torch.zeros(5, n_pad, 72, device=dev)
        g_emb_soa = torch.empty(20, n, device=dev)
        g_latent = torch.zeros(8, device=dev)
        out = _abi.InvrMlpBwdOut()
        out.g_emb, out.gz, out.a, out.n_pad, out.g_latent = g_emb_soa.data_ptr(), gz.data_ptr(), a.data_ptr(), n_pad, g_latent.data_ptr()
        g_raw = g_raw.to(torch.float32).contiguous()
        _abi.check(_abi.lib().invr_part_mlp_bwd(C.byref(ctx.model), ctx.pid, _abi.ptr(li, torch.int64), _abi.ptr(emb_soa), _abi.ptr(dirs_soa),
                                                n, _abi.ptr(g_raw), C.byref(out), _abi.stream_ptr()))
        # all weight gradients of the part in ONE slab-batched GEMM: (5 S, 64, 2048) x (5 S, 2048, 72), summed over the S slabs
        S = n_pad // _SPLIT_ROWS
        dW = torch.bmm(gz.view(5 * S, _SPLIT_ROWS, 64).transpose(1, 2), a.view(5 * S, _SPLIT_ROWS, 72)).view(5, S, 64, 72).sum(1)
        db = gz.sum(1)
        col = _slot_of_col(dev)
        # parameter order: occ W0 b0 W1 b1, rgb W0 b0 [W1 b1] Wout bout
        grads = [dW[0, :, :19], db[0], dW[1, :17, :64], db[1, :17], dW[2][:, col], db[2]]
        if three:
            grads += [dW[3, :, :64], db[3]]
        grads += [dW[4, :3, :64], db[4, :3]]
        g_emb = g_emb_soa[:19].t()
        g_lat = torch.zeros_like(rgb_latent)
        g_lat[li[0]] = g_latent
        return (g_emb, None, None, None, None, g_lat) + tuple(grads)


def part_field_hip(pn, tpts, tdirs, latent_index, model, pid):
    """part_field with the MLPs on the HIP forward / backward kernels."""
    e = pn.embedder
    emb = GridEncodeFn.apply(tpts, e.dense if e.separate_dense else None, e.hash, e.bounds, e.spec)
    params = []
    for mlp in (pn.occ, pn.rgb):
        for l in mlp.linears:
            params += [l.weight, l.bias]
    return PartMlpFn.apply(emb, tdirs, model, pid, latent_index, pn.rgb_latent, *params)


def part_field(pn, tpts, tdirs, latent_index, n_freq):
    """part_base_network.Network.forward (:44-63) with gradients (torch MLPs; the reference path of the tests)."""
    e = pn.embedder
    emb = GridEncodeFn.apply(tpts, e.dense if e.separate_dense else None, e.hash, e.bounds, e.spec)
    return part_mlps_torch(pn, emb, tdirs, latent_index, n_freq)


def part_mlps_torch(pn, emb, tdirs, latent_index, n_freq):
    """The two MLPs of a part on an embedding (n,19), torch ops (part_base_network.py:50-63)."""
    h = mlp_forward(pn.occ, emb)
    occ = 1 - torch.exp(-F.softplus(h[..., :1]))
    lat = pn.rgb_latent[latent_index.reshape(-1)[0]][None].expand(emb.shape[0], -1)
    x = torch.cat([emb, dir_encode(tdirs, n_freq), h[..., 1:], lat], -1)
    rgb = torch.sigmoid(mlp_forward(pn.rgb, x))
    return torch.cat([rgb, occ], -1)


def distortion(weights, z):
    """inb_renderer.py:96-103."""
    nz = torch.cat([z[:, 1:], z[:, -1:]], -1)
    mid = (z + nz) / 2
    return ((weights[:, :, None] * weights[:, None, :]) * (mid[:, :, None] - mid[:, None, :]).abs()).sum(-1).sum(-1)


def render_train(net, batch, geo, views, stats, n_rays, S, pair_noise, ctx=None, rays=None, jitter=None):
    """Differentiable recomputation of one training forward on the pair lists `views` left by the HIP
    geometry pass `geo` (= Network.render_rays output).  Returns the reference's train-mode dict."""
    cfg = net.cfg
    P = NUM_PARTS
    dev = geo['z_vals'].device
    Na, cap = int(stats[0]), views['cap']
    n_freq = cfg.viewdir_embedder.kwargs['res']
    cnts = [int(stats[1 + p]) for p in range(P)]
    xb, dirs, rows = [], [], []
    for p in range(P):
        c = cnts[p]
        r = views['l_r'][p][:, :c].t()
        xb.append((views['l_x'][p][:, :c].t() - r).contiguous())                       # init_bigpose (no grad)
        dirs.append(views['l_d'][p][:, :c].t().contiguous())
        slots = views['l_slot'][p][:c].long()
        rows.append(torch.where(slots == cap, torch.full_like(slots, Na), slots))   # the far constant -> row Na
    resd_all = deform(net, batch, torch.cat(xb, 0))
    resd_p = torch.split(resd_all, cnts, 0)
    far = views['farflags'][:Na].to(torch.int32)
    hip_mlp = cfg.get('train_hip_mlp', True)
    if hip_mlp:
        keep_model = []
        model = _abi.make_model({k: v for k, v in net.named_parameters()}, cfg, keep_model)
    raws_flat = torch.zeros((Na + 1) * P, 4, device=dev)
    resd_flat = torch.zeros((Na + 1) * P, 3, device=dev)
    tpts_flat = torch.zeros((Na + 1) * P, 3, device=dev)
    for p in range(P):
        if cnts[p] == 0:
            continue
        pn = net.tpose_human.part_networks[p]
        tpose = xb[p] + resd_p[p]                                                        # :111
        raw = (part_field_hip(pn, tpose, dirs[p], batch['latent_index'], model, p) if hip_mlp
               else part_field(pn, tpose, dirs[p], batch['latent_index'], n_freq))
        flat = rows[p] * P + p
        raws_flat = raws_flat.index_copy(0, flat, raw)
        resd_flat = resd_flat.index_copy(0, flat, resd_p[p])
        tpts_flat = tpts_flat.index_copy(0, flat, xb[p])
        fr = ((far >> p) & 1).nonzero(as_tuple=True)[0]
        if fr.numel():                                   # far pairs share the part constant (last list entry)
            ff = fr * P + p
            raws_flat = raws_flat.index_copy(0, ff, raw[-1:].expand(fr.numel(), 4))
            resd_flat = resd_flat.index_copy(0, ff, resd_p[p][-1:].expand(fr.numel(), 3))
            tpts_flat = tpts_flat.index_copy(0, ff, xb[p][-1:].expand(fr.numel(), 3))
    raws = raws_flat.view(Na + 1, P, 4)[:Na]
    resd = resd_flat.view(Na + 1, P, 3)[:Na]
    tpts = tpts_flat.view(Na + 1, P, 3)[:Na]
    if ctx is not None and Na:
        # rows that are neither listed nor far: init_bigpose under that part's 4-NN weights, as the reference computes it for
        # all Na x P rows (inb_part_network_multiassign.py:96-120) — see renderer.dense_train_rows
        from . import stages
        ro, rd, nr, fa = rays
        with torch.no_grad():
            pts, pdirs = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, views['active_idx'][:Na], jitter=jitter)
            bw, _ = stages.knn_blend(ctx.scene, pts)
            tp, _, _ = stages.warp_deform(ctx.scene, ctx.model, pts, pdirs, bw, torch.zeros(Na, P, dtype=torch.uint8, device=dev))
            known = (((views['pflags'][:Na].to(torch.int32) | far)[:, None] >> torch.arange(P, device=dev)[None]) & 1).bool()
        tpts = torch.where(known[..., None], tpts, tp)
    tocc = raws[..., 3]
    aggr = cfg.get('aggr', '') or ''
    if aggr == 'mean':
        merged = raws.mean(dim=1)                                                        # :236-239
    elif aggr in ('dist', 'mindist'):                                                    # :240-251: part_dist of every (survivor, part)
        from . import stages
        ro, rd, nr, fa = rays
        with torch.no_grad():
            ppts, _ = stages.pose_points(ctx.scene, ro, rd, nr, fa, S, views['active_idx'][:Na], jitter=jitter, want_dirs=False)
            _, pdist = stages.knn_blend(ctx.scene, ppts)
        if aggr == 'dist':
            merged = torch.sum(raws * torch.nn.functional.normalize(1.0 / (pdist + 1e-5), dim=-1)[..., None], dim=1)
        else:
            merged = raws[torch.arange(Na, device=dev), pdist.argmin(dim=1)]
    else:
        ind = tocc.argmax(dim=1)                                                         # :253
        merged = raws[torch.arange(Na, device=dev), ind]
    act = views['active_idx'][:Na].long()
    raw_full = torch.zeros(n_rays * S, 4, device=dev).index_copy(0, act, merged)        # :156-159
    weights, rgb_map, acc_map = CompositeFn.apply(raw_full.view(n_rays, S, 4))
    ret = {'rgb_map': rgb_map[None], 'acc_map': acc_map[None], 'raw': raw_full[None], 'occ': raw_full[None, :, 3:],
           'resd': resd.reshape(1, -1, 3), 'tpts': tpts.reshape(1, -1, 3).detach(), 'tocc': tocc.reshape(1, -1, 1)}
    if cfg.use_pair_reg:                                                                 # inb_renderer.py:78-94
        reg = ((tocc.detach().reshape(-1) - 0.5).abs() < 0.02).nonzero(as_tuple=True)[0]
        if reg.numel():
            reg_tpts = ret['tpts'].reshape(-1, 3)[reg]
            reg_resd = resd.reshape(-1, 3)[reg]
            neighbor = reg_tpts + (pair_noise(reg_tpts[None])[0] - 0.5) * 0.01
            nei = deform(net, batch, neighbor)
            ret['oresd'] = torch.cat([reg_resd[None], nei[None]], dim=1)
        else:
            ret['oresd'] = torch.zeros(1, 0, 3, device=dev)
    if cfg.use_reg_distortion:
        ret['reg_distortion_loss'] = distortion(weights, geo['z_vals'])[None]
    return ret


# ---------------------------------------------------------------------------------------------------------------------
# Fused training iteration: invr_train_fwd / invr_train_bwd (csrc/k_train.hip).  No host round trip, no torch op soup.
# ---------------------------------------------------------------------------------------------------------------------
TERM_OFFSET_SUM, TERM_OFFSET_ROWS, TERM_PAIR_SUM, TERM_PAIR_ROWS, TERM_LEN = 0, 1, 2, 3, 8


class GradArena:
    """Persistent gradient storage of one Network for the fused training path.

    One flat fp32 buffer holds the gradients of every small parameter (MLPs, latent codes, deformer tables) as views
    (`p.grad` aliases them); the five part grids keep compact row-scalar gradients (Embedder.row_grad).  The fused backward
    ACCUMULATES into the arena; `zero()` (FusedAdam.zero_grad / after a step) clears it with a handful of memsets.  Because
    the addresses never change, the InvrTrainGrads struct and FusedAdam's device tensor table are built once."""

    def __init__(self, net):
        self.net = net
        dev = next(net.parameters()).device
        self.tables = []                                      # part-grid tables: no dense gradient in this mode
        for pn in net.tpose_human.part_networks:
            e = pn.embedder
            self.tables += [e.hash] + ([e.dense] if e.separate_dense else [])
        tab_ids = {id(t) for t in self.tables}
        self.small = [p for p in net.parameters() if p.requires_grad and id(p) not in tab_ids]
        n = sum(p.numel() for p in self.small)
        self.embedders = [pn.embedder for pn in net.tpose_human.part_networks]
        # ONE allocation for the flat buffer and the five row-scalar gradients (round 6): zero() is one fill instead of six launches in
        # front of every backward.  The store is kept on the network, so a throw-away arena (backward without FusedAdam) does not
        # allocate 68 MB per iteration either.
        keep = []
        sizes = [int(_abi.lib().invr_grid_row_sums_len(C.byref(e.grid_struct(keep)))) for e in self.embedders]
        pad = lambda k: (k + 63) // 64 * 64
        key = (n, tuple(sizes), str(dev))
        store = getattr(net, '_grad_store', None)
        if store is None or store[0] != key:
            store = (key, torch.zeros(pad(n) + sum(pad(k) for k in sizes), device=dev, dtype=torch.float32))
            net._grad_store = store
        self.all = store[1]
        self.all.zero_()
        self.flat = self.all[:n]
        self.views, o = {}, 0
        for p in self.small:
            self.views[id(p)] = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        o = pad(n)
        for e, k in zip(self.embedders, sizes):
            e._row_grad = self.all[o:o + k]
            e.row_grad_dirty = False
            o += pad(k)
        self.struct = self._build()
        self.dirty = False

    def grad_of(self, p):
        return self.views[id(p)]

    def _build(self):
        net, G = self.net, _abi.InvrTrainGrads()
        g = lambda p: self.views[id(p)].data_ptr()
        for i, pn in enumerate(net.tpose_human.part_networks):
            P = G.part[i]
            P.row_grad = pn.embedder.row_grad().data_ptr()
            for k, l in enumerate(pn.occ.linears):
                P.occ_w[k], P.occ_b[k] = g(l.weight), g(l.bias)
            for k, l in enumerate(pn.rgb.linears):
                P.rgb_w[k], P.rgb_b[k] = g(l.weight), g(l.bias)
            P.rgb_latent = g(pn.rgb_latent)
        e = net.tpose_deformer.embedder
        G.deform_hash = g(e.hash)
        if e.separate_dense:
            G.deform_dense = g(e.dense)
        for k, idx in enumerate((0, 2, 4)):
            G.deform_w[k], G.deform_b[k] = g(net.tpose_deformer.mlp[idx].weight), g(net.tpose_deformer.mlp[idx].bias)
        G.part_active = None      # every tensor takes every Adam step: the reference runs all five part networks even on zero points
                                  # (inb_part_network_multiassign.py:223-229), so their gradients are zeros, never None
        return G

    def publish(self):
        """p.grad of every small parameter = its arena view (aliases, no copy); table parameters keep grad None."""
        for p in self.small:
            if p.grad is not self.views[id(p)]:
                p.grad = self.views[id(p)]
        for e in self.embedders:
            e.row_grad_dirty = True
        self.dirty = True

    def zero(self):
        if all(e._row_grad.data_ptr() >= self.all.data_ptr() and e._row_grad.data_ptr() < self.all.data_ptr() + self.all.numel() * 4 for e in self.embedders):
            self.all.zero_()
        else:                                  # (an embedder re-created its gradient, e.g. after a device move)
            self.flat.zero_()
            for e in self.embedders:
                e.row_grad().zero_()
        for e in self.embedders:
            e.row_grad_dirty = False
        self.dirty = False

    def expand_tables(self):
        """Dense .grad of the part tables from the row-scalar gradients (tests, foreign optimisers)."""
        for e in self.embedders:
            e.expand_row_grad()


def check_workspace(net, ws, gen, what):
    """The fused training forward leaves its pair lists and activations in the network's shared workspace; the backward and the
    lazily materialised tensors read them again.  Any library call on the same network in between (a second training forward
    — gradient accumulation, summed losses —, an eval render, Network.forward, geometry_pass) overwrites them: raise instead of
    differentiating someone else's pair lists."""
    if net._ws is not ws or getattr(net, '_ws_gen', None) != gen:
        raise RuntimeError('invr: the training %s needs the workspace of ITS forward, but another call on this network has used the '
                           'workspace since (a second forward before the backward, an eval render, Network.forward ...).  Run '
                           'backward() / read resd, tpts, tocc, oresd before the next call on the network.' % what)


class TrainRenderFn(torch.autograd.Function):
    """Renderer.render in train mode + the regulariser reductions as ONE differentiable node: forward = invr_train_fwd,
    backward = invr_train_bwd.  `params` are the network's parameters (listed so that autograd connects the node to them).
    Gradient delivery: with a GradArena (fused mode, FusedAdam) the backward accumulates into the arena and returns None for
    the parameters; without one it returns fresh dense gradients like any autograd node (any optimiser, DDP hooks)."""

    @staticmethod
    def forward(ctx, net, rctx, arena, ray_o, ray_d, near, far, n_samples, jitter, pair_noise, max_active, *params):
        L = _abi.lib()
        dev = ray_o.device
        f = lambda t: t.detach().to(torch.float32).contiguous()
        ray_o, ray_d, near, far = f(ray_o), f(ray_d), f(near), f(far)
        n, S = ray_o.shape[0], int(n_samples)
        N = n * S
        rgb = torch.empty(n, 3, device=dev); acc = torch.empty(n, device=dev)
        raw = torch.empty(N, 4, device=dev); occ = torch.empty(N, device=dev)
        weights = torch.empty(n, S, device=dev); z = torch.empty(n, S, device=dev)
        dist = torch.empty(n, device=dev)
        terms = torch.empty(TERM_LEN, device=dev)
        stats = torch.zeros(_abi.STATS_LEN, dtype=torch.int32, device=dev)
        nbytes = L.invr_train_workspace_bytes(n, S, max_active)
        ws = net.workspace(nbytes, dev)
        ctx.ws_gen = net._ws_gen
        jit = None if jitter is None else f(jitter)
        noise = None if pair_noise is None else f(pair_noise)
        _abi.check(L.invr_train_fwd(C.byref(rctx.scene), C.byref(rctx.model), _abi.ptr(ray_o), _abi.ptr(ray_d), _abi.ptr(near), _abi.ptr(far),
                                    _abi.ptr(jit), n, S, _abi.ptr(noise), 0 if noise is None else noise.shape[0],
                                    _abi.ptr(rgb), _abi.ptr(acc), _abi.ptr(raw), _abi.ptr(occ), _abi.ptr(weights), _abi.ptr(z),
                                    _abi.ptr(dist), _abi.ptr(terms), _abi.ptr(stats, torch.int32),
                                    C.c_void_p(ws.data_ptr()), nbytes, max_active, _abi.stream_ptr()))
        ctx.net, ctx.rctx, ctx.arena, ctx.ws, ctx.dims = net, rctx, arena, ws, (n, S, max_active, nbytes)
        ctx.params = params
        ctx.save_for_backward(raw, weights, z)
        ctx.mark_non_differentiable(occ, weights, z, stats)
        return rgb, acc, raw, dist, terms, occ, weights, z, stats

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_raw, g_dist, g_terms, *_unused):
        L = _abi.lib()
        raw, weights, z = ctx.saved_tensors
        net, rctx, arena = ctx.net, ctx.rctx, ctx.arena
        check_workspace(net, ctx.ws, ctx.ws_gen, 'backward')
        n, S, max_active, nbytes = ctx.dims
        dev = raw.device
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        g_rgb = c(g_rgb) if g_rgb is not None else torch.zeros(n, 3, device=dev)
        g_terms = c(g_terms)
        g_off = g_pair = None
        if g_terms is not None:
            g_off, g_pair = g_terms[TERM_OFFSET_SUM:TERM_OFFSET_SUM + 1], g_terms[TERM_PAIR_SUM:TERM_PAIR_SUM + 1]
        own = arena if arena is not None else GradArena(net)       # no arena: a throw-away one, handed to autograd below
        g_acc, g_dist, g_raw = c(g_acc), c(g_dist), c(g_raw)

        def run(stages):
            _abi.check(L.invr_train_bwd(C.byref(rctx.scene), C.byref(rctx.model), n, S, _abi.ptr(raw), _abi.ptr(weights), _abi.ptr(z),
                                        _abi.ptr(g_rgb), _abi.ptr(g_acc), _abi.ptr(g_dist), _abi.ptr(g_raw),
                                        C.c_void_p(g_off.data_ptr()) if g_off is not None else None,
                                        C.c_void_p(g_pair.data_ptr()) if g_pair is not None else None,
                                        C.byref(own.struct), stages, C.c_void_p(ctx.ws.data_ptr()), nbytes, max_active, _abi.stream_ptr()))
        reducer = getattr(arena, 'reducer', None) if arena is not None else None
        if reducer is None:
            run(_abi.BWD_ALL)
        else:
            # data parallel: the five parts' chains run side by side inside ONE call (library-owned streams); their row-scalar
            # gradients — the large blocks, largest first — are then all-reduced on the collective's own stream beside the
            # deformer stage of the backward, the small tensors after it (dist_train.GradReducer)
            run(_abi.BWD_ALL & ~_abi.BWD_DEFORMER)
            for p in reducer.part_order:
                reducer.reduce_part(p)
            run(_abi.BWD_DEFORMER)
            reducer.reduce_small()
        head = (None,) * 11
        if arena is not None:
            arena.publish()
            return head + (None,) * len(ctx.params)
        # standard autograd delivery: dense gradients (the part tables expanded from their row scalars)
        dense = {}
        for e in own.embedders:
            gd, gh = e.expand_row_grad()
            e.hash.grad = None
            dense[id(e.hash)] = gh
            if e.separate_dense:
                e.dense.grad = None
                dense[id(e.dense)] = gd
            e.row_grad().zero_()
            e.row_grad_dirty = False
        out = []
        for p in ctx.params:
            if id(p) in dense:
                out.append(dense[id(p)])
            elif id(p) in own.views:
                out.append(own.views[id(p)].clone())
            else:
                out.append(None)
        return head + tuple(out)


class TrainLossFn(torch.autograd.Function):
    """NetworkWrapper's objective on the fused node's outputs as ONE node (invr_train_loss_fwd / _bwd): rgb_map (n,3), the target
    (n,3), the distortion regulariser (n,) or None and the node's `terms` -> (out8 = [loss, img_loss, psnr, reg_dist, offset_loss,
    pair_loss, 0, 0], err (n,)); only out8[0] carries a gradient.  As torch ops the same arithmetic was ~35 tiny kernels and as many
    again in their backward, 1 ms of host-bound time per iteration."""

    @staticmethod
    def forward(ctx, rgb, gt, dist, terms, w_pair, w_dist, w_off, use_pair):
        L = _abi.lib()
        rgb, gt = rgb.contiguous(), gt.contiguous().to(torch.float32)
        n = rgb.shape[0]
        out = torch.empty(8, device=rgb.device)
        err = torch.empty(n, device=rgb.device)
        dist_c = dist.contiguous() if dist is not None else None
        _abi.check(L.invr_train_loss_fwd(_abi.ptr(rgb), _abi.ptr(gt), _abi.ptr(dist_c), _abi.ptr(terms), n, w_pair, w_dist, w_off, int(use_pair),
                                         _abi.ptr(out), _abi.ptr(err), _abi.stream_ptr()))
        ctx.save_for_backward(rgb, gt, terms)
        ctx.has_dist, ctx.w = dist is not None, (w_pair, w_dist, w_off, int(use_pair))
        ctx.mark_non_differentiable(err)
        return out, err

    @staticmethod
    def backward(ctx, g_out, _g_err):
        L = _abi.lib()
        rgb, gt, terms = ctx.saved_tensors
        n = rgb.shape[0]
        g_loss = g_out[:1].contiguous()
        g_rgb = torch.empty_like(rgb)
        g_dist = torch.empty(n, device=rgb.device) if ctx.has_dist else None
        g_terms = torch.empty(TERM_LEN, device=rgb.device)
        w_pair, w_dist, w_off, use_pair = ctx.w
        _abi.check(L.invr_train_loss_bwd(_abi.ptr(rgb), _abi.ptr(gt), _abi.ptr(terms), n, w_pair, w_dist, w_off, use_pair, _abi.ptr(g_loss),
                                         _abi.ptr(g_rgb), _abi.ptr(g_dist), _abi.ptr(g_terms), _abi.stream_ptr()))
        return g_rgb, None, g_dist, g_terms, None, None, None, None


class LazyTrainRet(dict):
    """The train-mode return dict of Renderer.render.  rgb_map / acc_map / raw / occ / reg_distortion_loss and the fused
    regulariser terms (offset_loss, pair_loss: differentiable scalars) are present; the reference's dynamic-shape tensors
    resd / tpts / tocc / oresd are materialised from the workspace on first access (that read-back synchronises with the
    device, like the reference's own nonzero()s; they are detached — the gradient flows through the fused terms)."""
    def __init__(self, base, lazy_keys, materialise, thunks=None):
        super().__init__(base)
        self._lazy, self._mat, self._done = tuple(lazy_keys), materialise, False
        self._thunks = dict(thunks or {})       # cheap derived entries (a few torch ops, no synchronisation), built on first access:
                                                # the trainer's fused objective never touches them (offset_loss / pair_loss)

    def _thunk(self, k):
        if k in self._thunks:
            dict.__setitem__(self, k, self._thunks.pop(k)())

    def _fill(self):
        for k in tuple(self._thunks):
            self._thunk(k)
        if not self._done:
            self._done = True
            for k, v in self._mat().items():
                dict.__setitem__(self, k, v)

    def __contains__(self, k):
        return dict.__contains__(self, k) or k in self._thunks or (not self._done and k in self._lazy)

    def __getitem__(self, k):
        self._thunk(k)
        if not dict.__contains__(self, k) and k in self._lazy:
            self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        return self[k] if k in self else d

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    # every other dict view fills first too (dict's C fast paths — values(), iteration, len(), dict(ret), copy() — would not see the
    # thunks / lazy tensors otherwise: a host loop `for k, v in dict(ret).items()` must find offset_loss / pair_loss / resd ...)
    def values(self):
        self._fill()
        return dict.values(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._thunks) + (0 if self._done else sum(1 for k in self._lazy if not dict.__contains__(self, k)))

    def copy(self):
        self._fill()
        return dict(self)

    def pop(self, k, *default):
        if k in self:
            self[k]
        return dict.pop(self, k, *default)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None

    def __reduce__(self):
        self._fill()
        return (dict, (dict(self),))
