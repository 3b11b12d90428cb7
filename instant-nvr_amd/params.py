"""Model geometry (hash-grid level tables) and checkpoint layout.

``grid_spec`` restates the constructor arithmetic of the reference's multi-resolution
encoder (lib/networks/embedders/part_base_embedder.py:13-104) — per-level resolution,
float32 cell size, dense/hash split, prime table length — and ``init_state_dict`` emits a
seeded parameter set under the reference's exact ``state_dict`` key names and shapes
(SURVEY.md §5 "Checkpoint / resume") so ``.pth`` files interchange.
"""
from collections import OrderedDict

import numpy as np

from .config import PART_NAMES, NUM_PARTS

CORNER_OFFSETS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
                           [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], dtype=np.float32)  # embedder :81-88


def next_prime(n):
    """Smallest prime > n (the reference calls sympy.nextprime, part_base_embedder.py:42)."""
    def is_prime(m):
        if m < 2:
            return False
        if m % 2 == 0:
            return m == 2
        i = 3
        while i * i <= m:
            if m % i == 0:
                return False
            i += 2
        return True
    m = n + 1
    while not is_prime(m):
        m += 1
    return m


def grid_spec(n_levels=16, n_features_per_level=16, b=1.38, log2_hashmap_size=18, base_resolution=2,
              sum=True, sum_over_features=True, separate_dense=True, use_batch_bounds=True,
              include_input=True, bbox=((0, 0, 0), (1, 1, 1)), **_unused):
    L, F = int(n_levels), int(n_features_per_level)
    T = next_prime(2 ** int(log2_hashmap_size))
    res = [int(base_resolution * b ** i) for i in range(L)]                      # :52
    cnt = [r ** 3 for r in res]                                                  # :53
    size = np.array([1 / (r - 1) for r in res]).astype(np.float32)               # :54,57 (float32 tensor)
    start_hash = L
    for i in range(L):
        if cnt[i] > T:                                                           # :63-68
            start_hash = i
            break
    sep = bool(separate_dense and start_hash)                                    # :69
    csum = np.cumsum(np.array(cnt, dtype=np.int64))
    dense_off = [0] + [int(c) for c in csum[:-1]]                                # :129 (entries_sum shift)
    out_dim = (L if sum_over_features else F) if sum else L * F
    out_dim += 3 if include_input else 0
    return dict(L=L, F=F, T=T, res=res, cnt=cnt, size=size, start_hash=start_hash, separate_dense=sep,
                dense_rows=int(csum[start_hash - 1]) if sep else 0, dense_off=dense_off,
                n_hash=L - start_hash if sep else L, sum=bool(sum), sum_over_features=bool(sum_over_features),
                include_input=bool(include_input), use_batch_bounds=bool(use_batch_bounds),
                bbox=np.asarray(bbox, dtype=np.float32).reshape(2, 3), out_dim=out_dim)


def part_grid_spec(cfg, partname):
    pc = cfg.partnet[partname]
    return grid_spec(bbox=pc.bbox, **pc.embedder.kwargs)


def deformer_grid_spec(cfg):
    return grid_spec(**cfg.tpose_deformer.embedder.kwargs)


def mlp_dims(cfg, partname):
    """Layer widths of the two per-part MLPs (part_base_network.py:11-42)."""
    emb = part_grid_spec(cfg, partname)['out_dim']
    occ = cfg.network.occ
    occ_dims = [emb] + [occ['d_hidden']] * occ['n_layers'] + [1 + cfg.geo_feature_dim]
    vres = cfg.viewdir_embedder.kwargs['res']
    dir_dim = 3 + 3 * 2 * vres
    rgb_in = emb + dir_dim + cfg.geo_feature_dim + cfg.latent_code_dim
    ck = cfg.partnet[partname].color_network.kwargs
    rgb_dims = [rgb_in] + [ck['d_hidden']] * ck['n_layers'] + [3]
    return occ_dims, rgb_dims


def _embedder_entries(prefix, spec, rng, table_std, sd):
    import torch
    L, F, T = spec['L'], spec['F'], spec['T']
    sd[prefix + 'bounds'] = torch.from_numpy(spec['bbox'].copy())
    sd[prefix + 'entries_size'] = torch.from_numpy(spec['size'].copy())
    sd[prefix + 'entries_num'] = torch.tensor(spec['res'], dtype=torch.int64)
    sd[prefix + 'entries_min'] = torch.zeros(L, dtype=torch.int64)
    sd[prefix + 'entries_cnt'] = torch.tensor(spec['cnt'], dtype=torch.int64)
    sd[prefix + 'entries_sum'] = torch.tensor(spec['cnt'], dtype=torch.int64).cumsum(0)
    if spec['separate_dense']:
        sd[prefix + 'dense'] = torch.from_numpy((rng.standard_normal((spec['dense_rows'], F)) * table_std).astype(np.float32))
        sd[prefix + 'hash'] = torch.from_numpy((rng.standard_normal((spec['n_hash'], T, F)) * table_std).astype(np.float32))
    else:
        sd[prefix + 'hash'] = torch.from_numpy((rng.standard_normal((L, T, F)) * table_std).astype(np.float32))
    sd[prefix + 'offsets'] = torch.from_numpy(CORNER_OFFSETS.copy())


def _linear(prefix, n_in, n_out, rng, sd, gain=1.0):
    import torch
    k = gain / np.sqrt(n_in)
    sd[prefix + 'weight'] = torch.from_numpy(rng.uniform(-k, k, (n_out, n_in)).astype(np.float32))
    sd[prefix + 'bias'] = torch.from_numpy(rng.uniform(-k, k, (n_out,)).astype(np.float32))


def init_state_dict(cfg, seed=0, table_std=0.1, mlp_gain=1.0):
    """Seeded parameters under the reference ``Network.state_dict()`` key set.

    Key order follows module registration order in the reference (deformer first,
    then the five part networks: embedder, embedder_dir, occ, rgb_latent?, rgb).
    """
    import torch
    rng = np.random.RandomState(seed)
    sd = OrderedDict()
    dspec = deformer_grid_spec(cfg)
    _embedder_entries('tpose_deformer.embedder.', dspec, rng, table_std, sd)
    _linear('tpose_deformer.mlp.0.', dspec['out_dim'], 32, rng, sd, mlp_gain)
    _linear('tpose_deformer.mlp.2.', 32, 32, rng, sd, mlp_gain)
    _linear('tpose_deformer.mlp.4.', 32, 3, rng, sd, mlp_gain)
    vres = cfg.viewdir_embedder.kwargs['res']
    for i, name in enumerate(PART_NAMES):
        p = 'tpose_human.part_networks.%d.' % i
        sd[p + 'rgb_latent'] = torch.from_numpy(
            (rng.standard_normal((cfg.num_latent_code, cfg.latent_code_dim)) * np.sqrt(2.0 / cfg.latent_code_dim)).astype(np.float32))
        _embedder_entries(p + 'embedder.', part_grid_spec(cfg, name), rng, table_std, sd)
        fb = (2.0 ** np.linspace(0.0, vres - 1, vres)).astype(np.float32)
        sd[p + 'embedder_dir.embedder.freq_bands'] = torch.from_numpy(
            np.broadcast_to(fb[:, None, None], (vres, 2, 1)).copy())
        occ_dims, rgb_dims = mlp_dims(cfg, name)
        for li in range(len(occ_dims) - 1):
            _linear(p + 'occ.linears.%d.' % li, occ_dims[li], occ_dims[li + 1], rng, sd, mlp_gain)
        for li in range(len(rgb_dims) - 1):
            _linear(p + 'rgb.linears.%d.' % li, rgb_dims[li], rgb_dims[li + 1], rng, sd, mlp_gain)
    return sd


def n_parameters(sd):
    return int(sum(v.numel() for v in sd.values()))
