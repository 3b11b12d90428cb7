"""Hot-path configuration.

Only the flags the render path reads are kept (SURVEY.md §5 "Config / flags"):
they carry the reference's names and the values of configs/inb/inb_377.yaml
merged over lib/config/config.py defaults (reference: lib/config/config.py:10-300,
configs/inb/inb_377.yaml:18-165,277-281).  When this package is driven by the
reference's own train_net.py / run.py the reference's global ``cfg`` is adopted
instead (``adopt(host_cfg)``), so yaml / CLI overrides made there are honoured.
"""
import copy

PART_NAMES = ['body', 'leg', 'head', 'larm', 'rarm']          # blend_utils.py:17
NUM_PARTS = 5                                                 # blend_utils.py:9
PART_BW_MAP = {                                               # blend_utils.py:10-16
    'body': [14, 13, 9, 6, 3, 0],
    'leg': [1, 2, 4, 5, 7, 8, 10, 11],
    'head': [12, 15],
    'larm': [16, 18, 20, 22],
    'rarm': [17, 19, 21, 23],
}
HASH_PRIMES = (1, 19349663, 83492791)                         # config.py:17 (cfg.ps)


class Node(dict):
    """Attribute dict (yacs-CfgNode-like read access)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _node(x):
    if isinstance(x, dict):
        return Node({k: _node(v) for k, v in x.items()})
    return x


def _part(log2_T, base_res, bbox, color_layers):
    d = {
        'embedder': {'kwargs': dict(n_levels=16, n_features_per_level=16, log2_hashmap_size=log2_T,
                                    base_resolution=base_res, b=1.38, sum=True, sum_over_features=True,
                                    separate_dense=True, use_batch_bounds=True)},
        'bbox': bbox,
        # rgb MLP: body/head fall back to MLP defaults (d_hidden 64, n_layers 2); leg/arms n_layers 1
        # (make_network.py:75-99, inb_377.yaml:103-107,138-142,158-162)
        'color_network': {'kwargs': dict(d_hidden=64, n_layers=color_layers)},
    }
    return d


DEFAULTS = {
    'N_samples': 64,
    'perturb': 1,
    'chunk': 4096,
    'render_chunk': 4096,
    'smpl_thresh': 0.05,
    'aggr': '',
    'tpose_viewdir': True,
    'use_pair_reg': True,
    'use_reg_distortion': True,
    'random_bg': False,
    'latent_code_dim': 8,
    'geo_feature_dim': 16,
    'num_train_frame': 100,
    'num_latent_code': 100,
    'knn_k': 4,
    'use_knn': True,
    'part_deform': False,
    'use_amp': False,
    'use_batch_bounds': True,
    'bbox_overlap': 0.2,
    'pair_loss_weight': 10.0,
    'reg_dist_weight': 0.1,
    'resd_loss_weight': 0.1,
    'train_fused_loss': True,   # training: NetworkWrapper's objective on the fused node's outputs as one node (autograd.TrainLossFn); False: torch ops
    'train_fused': True,     # training: ONE differentiable node (invr_train_fwd / invr_train_bwd); False: op-by-op autograd graph (autograd.render_train)
    'train_hip_mlp': True,   # training: part MLPs forward + backward on the HIP kernels (False: torch ops, autograd.part_field)
    'eval_row_sums': True,   # eval-mode renders read the part grids through derived row-sum tables (invr_grid_row_sums)
    # image loss: configs/inb/inb_377.yaml sets use_lpips True (VGG19 from torchvision, absent on this image).  The stand-alone
    # default is the plain MSE; adopt() takes the host's value and NetworkWrapper raises if no perceptual loss can be had.
    'use_lpips': False, 'use_ssim': False, 'use_fourier': False, 'use_tv_image': False, 'vgg19_weights': None,
    'network': {'occ': {'d_hidden': 64, 'n_layers': 1}},
    'viewdir_embedder': {'kwargs': {'res': 4, 'input_dims': 3}},
    'tpose_deformer': {'embedder': {'kwargs': dict(
        n_levels=8, n_features_per_level=2, log2_hashmap_size=14, base_resolution=4, b=1.38,
        sum=False, sum_over_features=True, separate_dense=True, use_batch_bounds=False,
        include_input=True)}},
    'partnet': {
        'body': _part(20, 16, [[-1, -1.2, -0.34], [0.8, 0.7, 0.5]], 2),
        'leg': _part(20, 2, [[-1, -1.2, -0.34], [0.8, -0.3, 0.5]], 1),
        'head': _part(18, 2, [[-0.3, 0.3, -0.3], [0.3, 0.7, 0.3]], 2),
        'larm': _part(15, 2, [[0.2, 0, -0.2], [0.9, 0.35, 0.2]], 1),
        'rarm': _part(15, 2, [[-0.9, 0, -0.2], [-0.2, 0.35, 0.2]], 1),
    },
}


def make_cfg(**overrides):
    """Fresh config = inb_377 hot-path defaults + keyword overrides.

    ``table_log2`` (int) is a convenience override that caps every part's
    log2_hashmap_size (used by tests / golden fixtures for small tables).
    """
    d = copy.deepcopy(DEFAULTS)
    table_log2 = overrides.pop('table_log2', None)
    for k, v in overrides.items():
        d[k] = v
    c = _node(d)
    if table_log2 is not None:
        for p in PART_NAMES:
            kw = c.partnet[p].embedder.kwargs
            kw['log2_hashmap_size'] = min(kw['log2_hashmap_size'], table_log2)
    return c


# Switches of the reference's path that this build does not implement.  A value other than the one listed makes
# Network.__init__ / adopt() RAISE: the HIP path would otherwise silently render the default behaviour (arg-max merge, K = 4, no
# background, shared deformer) under a config that asks for something else.  (cfg.N_importance — 128 in inb_377.yaml — is read by
# nothing in the reference's lib/: it is ignored here as it is there.)
UNSUPPORTED = {
    'knn_k': (4, 'cfg.knn_k != 4 (blend_utils.py:732-763): the KNN / skinning kernels are built for K = 4'),
    'part_deform': (False, 'cfg.part_deform (inb_part_network_multiassign.py:72,110): the reference itself asserts it off on this path'),
    'tpose_viewdir': (True, 'cfg.tpose_viewdir = False: TPoseHuman.forward indexes the (Na,P,3) view directions per part; the reference cannot run it either'),
    'use_knn': (True, 'cfg.use_knn = False: Network.__init__ of the reference asserts it'),
    'use_amp': (False, 'cfg.use_amp: the HIP path computes in fp32 (the reference\'s 1e-4 parity bar); half-precision autocast is not built'),
}


def validate(c):
    """Raise ValueError if `c` asks for a reference switch this build does not implement (see UNSUPPORTED)."""
    for k, (want, why) in UNSUPPORTED.items():
        have = c.get(k, want) if hasattr(c, 'get') else getattr(c, k, want)
        if have is None:
            have = want
        if isinstance(want, bool):
            ok = bool(have) == want
        elif isinstance(want, int):
            ok = int(have) == want
        else:
            ok = have == want
        if not ok:
            raise ValueError('invr: unsupported configuration %s = %r — %s' % (k, have, why))
    # cfg.random_bg True: inb_renderer.py:72 hands it to volume_rendering as render_weights' EPSILON (= 1.0; no background is ever added:
    # net_utils.py:29-44's use_random_bg stays False) — built in the fused paths; the op-by-op training graph composites with epsilon 0
    get = (lambda k, d: c.get(k, d)) if hasattr(c, 'get') else (lambda k, d: getattr(c, k, d))
    # cfg.aggr (inb_part_network_multiassign.py:236-256): '' = the part of largest occupancy (every INB yaml), 'mean', and since round 5
    # 'dist' (parts weighted by normalize(1 / (part_dist + 1e-5))) and 'mindist' (the part of smallest part_dist; the breakpoint() in
    # front of it is disarmed by lib/config/config.py:328-332).  part_dist is the eps-normalised KNN distance, ~0 for FAR parts, so in
    # both far parts dominate — reproduced as the reference computes it.
    if (get('aggr', '') or '') not in ('', 'mean', 'dist', 'mindist'):
        raise ValueError("invr: unsupported configuration aggr = %r — the reference's merges are '', 'mean', 'dist', 'mindist' "
                         "(inb_part_network_multiassign.py:236-256)" % (get('aggr', ''),))
    if bool(get('random_bg', False)) and not bool(get('train_fused', True)):
        raise ValueError('invr: unsupported configuration random_bg = True with train_fused = False — the op-by-op training graph is built '
                         'for epsilon 0 only')
    return c


cfg = make_cfg()


def set_cfg(new_cfg):
    """Replace the process-global hot-path config in place (keeps identity)."""
    cfg.clear()
    cfg.update(new_cfg)
    return cfg


def adopt(host_cfg):
    """Adopt the reference's ``lib.config.cfg`` (a yacs CfgNode) when hosted by it."""
    def conv(x):
        if hasattr(x, 'items'):
            return {k: conv(v) for k, v in x.items()}
        return x
    src = conv(host_cfg)
    merged = copy.deepcopy(DEFAULTS)
    for k in merged:
        if k in src:
            merged[k] = src[k]
    for p in PART_NAMES:
        pn = merged['partnet'][p]
        if 'color_network' not in pn:
            pn['color_network'] = {'kwargs': dict(d_hidden=64, n_layers=2)}
    return set_cfg(validate(_node(merged)))
