"""ctypes binding of libinvr.so (include/invr.h).  PyTorch-ROCm tensors in, tensors out.

The library is the product path: if it is missing this module raises at import (no CPU
fallback).  ``lib()`` loads it lazily so that CPU-only host logic (config, params, scene,
state_dict handling) stays importable on a box without the .so or without a GPU.
"""
import ctypes as C
import os

import torch

from . import params
from .config import NUM_PARTS, PART_NAMES

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('INVR_LIB_PATH') or os.path.join(HERE, 'libinvr.so')      # (INVR_LIB_PATH: experiment builds)

MAX_LEVELS, MAX_LINEAR, STATS_LEN = 16, 4, 16
_f32p, _i32p, _i64p, _u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)


class InvrGrid(C.Structure):
    _fields_ = [('dense', C.c_void_p), ('hash', C.c_void_p), ('bounds', C.c_void_p),
                ('n_levels', C.c_int32), ('n_features', C.c_int32), ('start_hash', C.c_int32),
                ('separate_dense', C.c_int32), ('table_len', C.c_int64),
                ('res', C.c_int32 * MAX_LEVELS), ('cell', C.c_float * MAX_LEVELS),
                ('dense_off', C.c_int64 * MAX_LEVELS),
                ('sum', C.c_int32), ('sum_over_features', C.c_int32), ('include_input', C.c_int32),
                ('row_sums', C.c_void_p)]


class InvrMlpBwdOut(C.Structure):
    _fields_ = [('g_emb', C.c_void_p), ('gz', C.c_void_p), ('a', C.c_void_p), ('n_pad', C.c_int64), ('g_latent', C.c_void_p)]


class InvrAdamTensor(C.Structure):
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('numel', C.c_int64), ('lr', C.c_float), ('weight_decay', C.c_float), ('bc1', C.c_float), ('bc2_sqrt', C.c_float),
                ('active', C.c_void_p), ('grad_shift', C.c_int32), ('step', C.c_int32)]


class InvrPartGrads(C.Structure):
    _fields_ = [('row_grad', C.c_void_p), ('occ_w', C.c_void_p * 4), ('occ_b', C.c_void_p * 4), ('rgb_w', C.c_void_p * 4),
                ('rgb_b', C.c_void_p * 4), ('rgb_latent', C.c_void_p)]


class InvrTrainGrads(C.Structure):
    _fields_ = [('part', InvrPartGrads * NUM_PARTS), ('deform_dense', C.c_void_p), ('deform_hash', C.c_void_p),
                ('deform_w', C.c_void_p * 4), ('deform_b', C.c_void_p * 4), ('part_active', C.c_void_p)]


class InvrMlp(C.Structure):
    _fields_ = [('weight', C.c_void_p * MAX_LINEAR), ('bias', C.c_void_p * MAX_LINEAR),
                ('dims', C.c_int32 * (MAX_LINEAR + 1)), ('n_linear', C.c_int32)]


class InvrPart(C.Structure):
    _fields_ = [('grid', InvrGrid), ('occ', InvrMlp), ('rgb', InvrMlp), ('rgb_latent', C.c_void_p),
                ('latent_dim', C.c_int32), ('num_latent_code', C.c_int32)]


class InvrModel(C.Structure):
    _fields_ = [('part', InvrPart * NUM_PARTS), ('deform_grid', InvrGrid), ('deform_mlp', InvrMlp),
                ('n_dir_freq', C.c_int32), ('geo_feature_dim', C.c_int32)]


class InvrScene(C.Structure):
    _fields_ = [('R', C.c_void_p), ('Th', C.c_void_p), ('A', C.c_void_p), ('big_A', C.c_void_p),
                ('pbw', C.c_void_p), ('pbw_dims', C.c_int32 * 3), ('pbw_channels', C.c_int32),
                ('pbounds', C.c_void_p), ('tuv', C.c_void_p), ('tuv_dims', C.c_int32 * 3),
                ('tbounds', C.c_void_p), ('part_pts', C.c_void_p), ('part_pbw', C.c_void_p),
                ('lengths2', C.c_void_p), ('part_stride', C.c_int32), ('frame_dim', C.c_void_p),
                ('latent_index', C.c_void_p), ('smpl_thresh', C.c_float), ('tpose_viewdir', C.c_int32),
                ('composite_eps', C.c_float), ('aggr', C.c_int32)]


class InvrWsLayout(C.Structure):
    _fields_ = [('cap', C.c_int64), ('lcap', C.c_int64), ('counters', C.c_int64), ('active_idx', C.c_int64),
                ('word_off', C.c_int64), ('mask', C.c_int64), ('pflags', C.c_int64), ('farflags', C.c_int64),
                ('l_slot', C.c_int64 * NUM_PARTS), ('l_nn', C.c_int64 * NUM_PARTS), ('l_w', C.c_int64 * NUM_PARTS),
                ('l_x', C.c_int64 * NUM_PARTS), ('l_d', C.c_int64 * NUM_PARTS), ('l_r', C.c_int64 * NUM_PARTS),
                ('emb', C.c_int64 * NUM_PARTS), ('occp', C.c_int64 * NUM_PARTS), ('wl', C.c_int64 * NUM_PARTS),
                ('wcnt', C.c_int64), ('wsel', C.c_int64), ('rgbw', C.c_int64), ('n_groups', C.c_int64), ('knn_dfar2', C.c_int64),
                ('byte_off', C.c_int64)]


EXPORTS = ['invr_last_error', 'invr_version', 'invr_sizeof', 'invr_workspace_bytes', 'invr_render_fwd',
           'invr_grid_encode_fwd', 'invr_sample_volume', 'invr_knn_blend', 'invr_warp_deform',
           'invr_part_field_workspace', 'invr_part_field_fwd', 'invr_composite_fwd',
           'invr_profile_enable', 'invr_profile_read', 'invr_workspace_layout', 'invr_deform_fwd',
           'invr_distortion_fwd', 'invr_grid_encode_bwd', 'invr_composite_bwd',
           'invr_field_workspace_bytes', 'invr_field_fwd', 'invr_geometry_fwd', 'invr_generate_rays',
           'invr_rigid_transformation', 'invr_pack_parts', 'invr_grid_row_sums_len', 'invr_grid_row_sums', 'invr_adam_chunk_elems', 'invr_adam_step', 'invr_part_mlp_fwd', 'invr_part_mlp_bwd',
           'invr_knn_neighbors', 'invr_pose_points', 'invr_adam_advance', 'invr_train_workspace_bytes', 'invr_train_fwd',
           'invr_train_bwd', 'invr_expand_row_grad', 'invr_train_loss_fwd', 'invr_train_loss_bwd',
           'invr_part_encode_workspace', 'invr_part_encode_fwd']
ABI_VERSION = 2          # include/invr.h INVR_ABI_VERSION
BWD_HEAD, BWD_DEFORMER, BWD_ALL = 1, 64, 127
NUM_STAGES = 14
STAGE_NAMES = ['cull', 'knn', 'warp'] + ['encode_%d' % p for p in range(5)] + ['mlp_%d' % p for p in range(5)] + ['composite']

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libinvr.so is not built (%s); run `python -c "import __graft_entry__ as g; g.build()"`. '
                               'There is no CPU fallback for the render path.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.invr_last_error.restype = C.c_char_p
        L.invr_version.restype = C.c_int
        if L.invr_version() != ABI_VERSION:
            raise RuntimeError('libinvr.so speaks ABI version %d, this binding %d (include/invr.h INVR_ABI_VERSION): rebuild the library '
                               '(python -m invr.build)' % (L.invr_version(), ABI_VERSION))
        L.invr_sizeof.restype = C.c_size_t
        L.invr_sizeof.argtypes = [C.c_int32]
        for i, t in enumerate((InvrGrid, InvrMlp, InvrPart, InvrModel, InvrScene, InvrWsLayout, InvrMlpBwdOut, InvrAdamTensor, InvrTrainGrads)):
            if L.invr_sizeof(i) != C.sizeof(t):
                raise RuntimeError('libinvr ABI mismatch: struct %s is %d bytes in the library, %d in the binding'
                                   % (t.__name__, L.invr_sizeof(i), C.sizeof(t)))
        L.invr_workspace_bytes.restype = C.c_size_t
        L.invr_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int64]
        L.invr_part_field_workspace.restype = C.c_size_t
        L.invr_part_field_workspace.argtypes = [C.c_int64]
        vp = C.c_void_p
        L.invr_render_fwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, vp, vp, vp, vp, C.c_int64,
                                      C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int64, vp]
        L.invr_grid_encode_fwd.argtypes = [C.POINTER(InvrGrid), vp, C.c_int64, vp, vp]
        L.invr_sample_volume.argtypes = [vp, C.c_int32 * 3, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp]
        L.invr_knn_blend.argtypes = [C.POINTER(InvrScene), vp, C.c_int64, vp, vp, vp]
        L.invr_adam_advance.argtypes = [vp, C.c_int32, C.c_double, C.c_double, vp]
        L.invr_adam_advance.restype = C.c_int
        L.invr_train_workspace_bytes.restype = C.c_size_t
        L.invr_train_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int64]
        L.invr_train_fwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, vp, vp, vp, vp, C.c_int64, C.c_int32, vp, C.c_int64,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int64, vp]
        L.invr_train_fwd.restype = C.c_int
        L.invr_train_bwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     C.POINTER(InvrTrainGrads), C.c_int32, vp, C.c_size_t, C.c_int64, vp]
        L.invr_train_bwd.restype = C.c_int
        L.invr_expand_row_grad.argtypes = [C.POINTER(InvrGrid), vp, vp, vp, vp]
        L.invr_expand_row_grad.restype = C.c_int
        L.invr_train_loss_fwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, vp, vp, vp]
        L.invr_train_loss_bwd.argtypes = [vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, vp, vp, vp, vp, vp]
        L.invr_train_loss_fwd.restype = L.invr_train_loss_bwd.restype = C.c_int
        L.invr_knn_neighbors.argtypes = [C.POINTER(InvrScene), vp, C.c_int64, vp, vp, vp, vp, vp]
        L.invr_knn_neighbors.restype = C.c_int
        L.invr_pose_points.argtypes = [C.POINTER(InvrScene), vp, vp, vp, vp, vp, C.c_int64, C.c_int32, vp, C.c_int64, vp, vp, vp]
        L.invr_pose_points.restype = C.c_int
        L.invr_warp_deform.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp]
        L.invr_part_field_fwd.argtypes = [C.POINTER(InvrModel), C.c_int32, vp, vp, vp, C.c_int64, vp, vp, C.c_size_t, vp]
        L.invr_part_encode_workspace.restype = C.c_size_t
        L.invr_part_encode_workspace.argtypes = [C.c_int64]
        L.invr_part_encode_fwd.argtypes = [C.POINTER(InvrGrid), vp, C.c_int64, C.c_int32, vp, vp, C.c_size_t, vp]
        L.invr_composite_fwd.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp]
        L.invr_workspace_layout.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.POINTER(InvrWsLayout)]
        L.invr_deform_fwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, C.c_int64, vp, vp]
        L.invr_distortion_fwd.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp]
        L.invr_grid_encode_bwd.argtypes = [C.POINTER(InvrGrid), vp, vp, C.c_int64, vp, vp, vp, vp]
        L.invr_composite_bwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, vp, vp]
        for n in ('invr_workspace_layout', 'invr_deform_fwd', 'invr_distortion_fwd', 'invr_grid_encode_bwd', 'invr_composite_bwd'):
            getattr(L, n).restype = C.c_int
        L.invr_geometry_fwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, vp, vp, vp, vp, C.c_int64, C.c_int32,
                                        vp, vp, vp, C.c_size_t, C.c_int64, vp]
        L.invr_geometry_fwd.restype = C.c_int
        dp = C.POINTER(C.c_double)
        L.invr_generate_rays.argtypes = [dp, dp, dp, dp, C.POINTER(C.c_float), C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
        L.invr_generate_rays.restype = C.c_int
        L.invr_grid_row_sums_len.argtypes = [vp]
        L.invr_grid_row_sums_len.restype = C.c_int64
        L.invr_grid_row_sums.argtypes = [vp, vp, vp]
        L.invr_grid_row_sums.restype = C.c_int
        L.invr_part_mlp_fwd.argtypes = [C.POINTER(InvrModel), C.c_int32, vp, vp, vp, C.c_int64, vp, vp, vp]
        L.invr_part_mlp_fwd.restype = C.c_int
        L.invr_part_mlp_bwd.argtypes = [C.POINTER(InvrModel), C.c_int32, vp, vp, vp, C.c_int64, vp, C.POINTER(InvrMlpBwdOut), vp]
        L.invr_part_mlp_bwd.restype = C.c_int
        L.invr_adam_chunk_elems.restype = C.c_int32
        L.invr_adam_step.argtypes = [vp, vp, vp, C.c_int64, C.c_double, C.c_double, C.c_float, vp]
        L.invr_adam_step.restype = C.c_int
        L.invr_rigid_transformation.argtypes = [vp, vp, vp, vp, vp]
        L.invr_rigid_transformation.restype = C.c_int
        L.invr_pack_parts.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, vp, vp, vp, vp]
        L.invr_pack_parts.restype = C.c_int
        L.invr_field_workspace_bytes.restype = C.c_size_t
        L.invr_field_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
        L.invr_field_fwd.argtypes = [C.POINTER(InvrScene), C.POINTER(InvrModel), vp, vp, C.c_int64, vp, vp, vp, vp, C.c_size_t, C.c_int64, vp]
        L.invr_field_fwd.restype = C.c_int
        L.invr_profile_enable.argtypes = [C.c_int32]
        L.invr_profile_read.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        for n in ('invr_render_fwd', 'invr_grid_encode_fwd', 'invr_sample_volume', 'invr_knn_blend',
                  'invr_warp_deform', 'invr_part_field_fwd', 'invr_composite_fwd'):
            getattr(L, n).restype = C.c_int
        _lib = L
    return _lib


def profile_enable(on):
    lib().invr_profile_enable(int(bool(on)))


def profile_read():
    """-> ({stage name: total ms since the last read}, number of renders)."""
    ms = (C.c_float * NUM_STAGES)()
    n = C.c_int32(0)
    check(lib().invr_profile_read(ms, C.byref(n)))
    return {STAGE_NAMES[i]: float(ms[i]) for i in range(NUM_STAGES)}, int(n.value)


def check(status):
    if status != 0:
        raise RuntimeError('libinvr: ' + lib().invr_last_error().decode())


def ws_views(ws, n_rays, S, max_active, n_active=None):
    """Zero-copy tensor views of the arrays invr_render_fwd left in the workspace `ws` (uint8 tensor)."""
    lay = InvrWsLayout()
    check(lib().invr_workspace_layout(n_rays, S, max_active, C.byref(lay)))
    lc = lay.lcap

    def view(off, count, dtype):
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return ws[off:off + nbytes].view(dtype)
    v = {'cap': lay.cap, 'lcap': lc,
         'counters': view(lay.counters, 16, torch.int32),
         'active_idx': view(lay.active_idx, lc, torch.int32),
         'word_off': view(lay.word_off, (n_rays * S + 1023) // 1024 * 16, torch.int32),
         'mask': view(lay.mask, (n_rays * S + 1023) // 1024 * 16, torch.int64),
         'byte_off': view(lay.byte_off, (n_rays * S + 1023) // 1024 * 128, torch.int32),        # (eval frames: depth-windowed order)
         'pflags': view(lay.pflags, lc, torch.uint8), 'farflags': view(lay.farflags, lc, torch.uint8),
         'knn_dfar2': view(lay.knn_dfar2, 1, torch.float32),
         'wsel': view(lay.wsel, lc, torch.uint8),                   # merge result per survivor (p / 8 + p / 255)
         'rgbw': view(lay.rgbw, (lc + 8) * 4, torch.float32).view(lc + 8, 4),      # winner's [rgb, occ] per slot; far constants at lc + p
         'wcnt': view(lay.wcnt, lay.n_groups * NUM_PARTS, torch.int32).view(lay.n_groups, NUM_PARTS)}
    v['occp'] = [view(lay.occp[p], lc, torch.float32) for p in range(NUM_PARTS)]      # occupancy of every listed pair
    v['wl'] = [view(lay.wl[p], lc, torch.int32) for p in range(NUM_PARTS)]
    for k in ('l_slot', 'l_x', 'l_d', 'l_r'):
        offs = getattr(lay, k)
        if k == 'l_slot':
            v[k] = [view(offs[p], lc, torch.int32) for p in range(NUM_PARTS)]
        else:
            v[k] = [view(offs[p], 3 * lc, torch.float32).view(3, lc) for p in range(NUM_PARTS)]
    v['l_nn'] = [view(lay.l_nn[p], 4 * lc, torch.int32).view(lc, 4) for p in range(NUM_PARTS)]      # neighbour rows inside part_pts[p]
    v['l_w'] = [view(lay.l_w[p], 4 * lc, torch.float32).view(lc, 4) for p in range(NUM_PARTS)]       # normalised gaussian weights
    v['emb'] = [view(lay.emb[p], 20 * lc, torch.float32).view(20, lc) for p in range(NUM_PARTS)]      # encoder outputs, SoA [k][pair]
    return v


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor of the expected dtype (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda, 'libinvr needs device tensors (got a CPU tensor)'
    assert t.dtype == dtype, (t.dtype, dtype)
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def make_grid(spec, dense, hsh, bounds, keep):
    """InvrGrid from params.grid_spec(...) + table tensors.  `keep` collects tensors to keep alive."""
    g = InvrGrid()
    hsh = _f32c(hsh); keep.append(hsh)
    g.hash = hsh.data_ptr()
    if dense is not None:
        dense = _f32c(dense); keep.append(dense)
        g.dense = dense.data_ptr()
    bounds = _f32c(bounds); keep.append(bounds)
    g.bounds = bounds.data_ptr()
    g.n_levels, g.n_features, g.start_hash = spec['L'], spec['F'], spec['start_hash']
    g.separate_dense, g.table_len = int(spec['separate_dense']), spec['T']
    for l in range(spec['L']):
        g.res[l] = spec['res'][l]
        g.cell[l] = float(spec['size'][l])
        g.dense_off[l] = spec['dense_off'][l]
    g.sum, g.sum_over_features, g.include_input = int(spec['sum']), int(spec['sum_over_features']), int(spec['include_input'])
    return g


def make_mlp(weights, biases, keep):
    m = InvrMlp()
    m.n_linear = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        w, b = _f32c(w), _f32c(b)
        keep += [w, b]
        m.weight[i], m.bias[i] = w.data_ptr(), b.data_ptr()
        m.dims[i], m.dims[i + 1] = w.shape[1], w.shape[0]
    return m


def make_model(sd, cfg, keep):
    """InvrModel from a reference-keyed state_dict of device tensors."""
    m = InvrModel()
    dspec = params.deformer_grid_spec(cfg)
    p = 'tpose_deformer.embedder.'
    m.deform_grid = make_grid(dspec, sd.get(p + 'dense'), sd[p + 'hash'], sd[p + 'bounds'], keep)
    m.deform_mlp = make_mlp([sd['tpose_deformer.mlp.%d.weight' % k] for k in (0, 2, 4)],
                            [sd['tpose_deformer.mlp.%d.bias' % k] for k in (0, 2, 4)], keep)
    for i, name in enumerate(PART_NAMES):
        q = 'tpose_human.part_networks.%d.' % i
        spec = params.part_grid_spec(cfg, name)
        part = m.part[i]
        part.grid = make_grid(spec, sd.get(q + 'embedder.dense'), sd[q + 'embedder.hash'], sd[q + 'embedder.bounds'], keep)
        occ_dims, rgb_dims = params.mlp_dims(cfg, name)
        part.occ = make_mlp([sd[q + 'occ.linears.%d.weight' % k] for k in range(len(occ_dims) - 1)],
                            [sd[q + 'occ.linears.%d.bias' % k] for k in range(len(occ_dims) - 1)], keep)
        part.rgb = make_mlp([sd[q + 'rgb.linears.%d.weight' % k] for k in range(len(rgb_dims) - 1)],
                            [sd[q + 'rgb.linears.%d.bias' % k] for k in range(len(rgb_dims) - 1)], keep)
        lat = _f32c(sd[q + 'rgb_latent']); keep.append(lat)
        part.rgb_latent = lat.data_ptr()
        part.latent_dim, part.num_latent_code = lat.shape[1], lat.shape[0]
    m.n_dir_freq = cfg.viewdir_embedder.kwargs['res']
    m.geo_feature_dim = cfg.geo_feature_dim
    return m


def make_scene(batch, cfg, keep):
    """InvrScene from the reference's collated batch dict (device tensors, leading dim 1)."""
    s = InvrScene()

    def f(k):
        t = _f32c(batch[k][0]); keep.append(t)
        return t
    s.R, s.Th, s.A, s.big_A = f('R').data_ptr(), f('Th').data_ptr(), f('A').data_ptr(), f('big_A').data_ptr()
    pbw, tuv = f('pbw'), f('tuv')
    s.pbw, s.tuv = pbw.data_ptr(), tuv.data_ptr()
    for a in range(3):
        s.pbw_dims[a], s.tuv_dims[a] = pbw.shape[a], tuv.shape[a]
    s.pbw_channels = pbw.shape[3]
    assert tuv.shape[3] == 2
    s.pbounds, s.tbounds = f('pbounds').data_ptr(), f('tbounds').data_ptr()
    pp, pb = f('part_pts'), f('part_pbw')
    assert pp.shape[0] == NUM_PARTS and pb.shape[2] == 24
    s.part_pts, s.part_pbw, s.part_stride = pp.data_ptr(), pb.data_ptr(), pp.shape[1]
    l2 = batch['lengths2'][0].to(torch.int64).contiguous(); keep.append(l2)
    s.lengths2 = l2.data_ptr()
    fd = batch['frame_dim'].reshape(-1)[:1].to(torch.float32).contiguous(); keep.append(fd)
    li = batch['latent_index'].reshape(-1)[:1].to(torch.int64).contiguous(); keep.append(li)
    s.frame_dim, s.latent_index = fd.data_ptr(), li.data_ptr()
    s.smpl_thresh, s.tpose_viewdir = float(cfg.smpl_thresh), int(bool(cfg.tpose_viewdir))
    s.aggr = {'': 0, 'mean': 1, 'dist': 2, 'mindist': 3}[cfg.get('aggr', '') or '']          # inb_part_network_multiassign.py:236-256 (INVR_AGGR_*)
    s.composite_eps = float(bool(cfg.get('random_bg', False)))          # inb_renderer.py:72 passes cfg.random_bg as render_weights' epsilon
    return s
