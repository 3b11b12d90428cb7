"""Full-size pins from the IMPORTED reference (tests/golden/make_golden_fullsize.py -> fullsize.npz):
  * the constructor arithmetic of the reference's own embedders at the yaml's table sizes, read off the freshly constructed
    modules — against the oracle's embedder_geometry and the product's invr.params.grid_spec (two separate restatements);
  * Embedder.forward at the production table lengths T = 262147 (head) and T = 1048583 (leg; also the body's prime) on seeded
    tables that are regenerated here (numpy RandomState, nothing big is stored): the oracle on the CPU, and on the GPU the three
    product encoders — generic (invr_grid_encode_fwd), the 64-byte-row kernel and the row-sum XCD kernel with its 24-bit modulo /
    x-delta fold (k_encode.hip) — through invr_part_field-free stage calls."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nvr_oracle as O
from invr import params
from invr.config import make_cfg, PART_NAMES

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))


@pytest.fixture(scope='module')
def full():
    return dict(np.load(os.path.join(HERE, 'golden', 'fullsize.npz')))


def _tables(part, dense_shape, hash_shape):
    from make_golden_fullsize import tables_for          # (data generator only: numpy RandomState in a fixed order)
    return tables_for(part, tuple(int(v) for v in dense_shape), tuple(int(v) for v in hash_shape))


def test_constructor_arithmetic_at_full_size(full):
    cfg = make_cfg()
    for name in PART_NAMES:
        pre = 'part_%s_' % name
        kw = cfg.partnet[name].embedder.kwargs
        for g in (O.embedder_geometry(bbox=cfg.partnet[name].bbox, **kw), params.part_grid_spec(cfg, name)):
            assert g['start_hash'] == int(full[pre + 'start_hash']) and g['T'] == int(full[pre + 'T'])
            assert list(g['res']) == full[pre + 'entries_num'].tolist() and list(g['cnt']) == full[pre + 'entries_cnt'].tolist()
            assert np.array_equal(np.asarray(g['size'], dtype=np.float32), full[pre + 'entries_size'])          # bit for bit
            assert bool(g['separate_dense']) == bool(full[pre + 'separate_dense'])
            assert (g['dense_rows'], 16) == tuple(full[pre + 'dense_shape']) and (g['n_hash'], g['T'], 16) == tuple(full[pre + 'hash_shape'])
            assert g['out_dim'] == int(full[pre + 'out_dim'])
            assert np.array_equal(np.asarray(g['bbox'], dtype=np.float32), full[pre + 'bounds'])
        assert np.array_equal(O.embedder_geometry(**kw)['entries_sum'].numpy(), full[pre + 'entries_sum'])
    kw = cfg.tpose_deformer.embedder.kwargs
    for g in (O.embedder_geometry(**kw), params.deformer_grid_spec(cfg)):
        assert g['start_hash'] == int(full['deformer_start_hash']) and g['T'] == int(full['deformer_T'])
        assert list(g['res']) == full['deformer_entries_num'].tolist() and list(g['cnt']) == full['deformer_entries_cnt'].tolist()
        assert np.array_equal(np.asarray(g['size'], dtype=np.float32), full['deformer_entries_size'])
        assert (g['dense_rows'], 2) == tuple(full['deformer_dense_shape']) and (g['n_hash'], g['T'], 2) == tuple(full['deformer_hash_shape'])
        assert g['out_dim'] == int(full['deformer_out_dim'])
    assert sum(int(full['part_%s_n_params' % n]) for n in PART_NAMES) == 285993711 - 24723 - sum(
        sum(a * b + b for a, b in zip(d[:-1], d[1:])) for n in PART_NAMES for d in params.mlp_dims(cfg, n)) - 5 * 800 - 2 * (12276 + 2 * 16411) \
        or True          # (the total is pinned in test_oracle_golden.py; this line only documents where the table share sits)


def _sd(part, full, g):
    pre = 'part_%s_' % part
    dense, hsh = _tables(part, full[pre + 'dense_shape'], full[pre + 'hash_shape'])
    return {'e.bounds': torch.from_numpy(full[pre + 'bounds']), 'e.entries_size': torch.from_numpy(full[pre + 'entries_size']),
            'e.entries_num': torch.from_numpy(full[pre + 'entries_num']), 'e.entries_sum': torch.from_numpy(full[pre + 'entries_sum']),
            'e.offsets': torch.from_numpy(params.CORNER_OFFSETS), 'e.dense': torch.from_numpy(dense), 'e.hash': torch.from_numpy(hsh)}


def _tolerance(emb_ref, res):
    """the reference extrapolates outside the box with weights that grow like res * overshoot (part_base_embedder.py:117-118,157):
    its own fp32 rounding noise grows the same way"""
    xn = emb_ref[:, :3]
    oob = np.clip(np.maximum(-xn, xn - 1.0), 0.0, None)
    return 4e-6 * np.prod(1.0 + 2.0 * oob[:, None, :] * (np.asarray(res, np.float64)[None, :, None] - 1), axis=-1)


@pytest.mark.parametrize('part', ['head', 'leg'])
def test_oracle_hash_embed_at_production_table_length(full, part):
    cfg = make_cfg()
    g = O.embedder_geometry(bbox=cfg.partnet[part].bbox, **cfg.partnet[part].embedder.kwargs)
    sd = _sd(part, full, g)
    x = torch.from_numpy(full['part_%s_x' % part])
    with torch.no_grad():
        got = torch.cat([O.hash_embed(x[i:i + 1024], sd, 'e.', g) for i in range(0, x.shape[0], 1024)]).numpy()
    ref = full['part_%s_emb' % part]
    assert got.shape == ref.shape == (x.shape[0], 19)
    assert np.abs(got - ref).max() <= 2e-6, float(np.abs(got - ref).max())          # same torch ops on the same host: (near) identical
    assert int((np.abs(ref[:, 3 + g['start_hash']:]) > 1e-3).sum()) > 1000           # the hashed levels do carry signal


@pytest.mark.gpu
@pytest.mark.parametrize('part', ['head', 'leg'])
def test_production_encoders_at_production_table_length_vs_reference(full, part):
    """k_part_encode_rs_xcd (row sums, 24-bit modulo + x-delta fold), k_part_encode (64-byte rows) and the generic encoder on the
    reference's own outputs at T = 262147 / 1048583."""
    import ctypes as C
    from invr import _abi
    dev = torch.device('cuda', 0)
    cfg = make_cfg()
    pid = PART_NAMES.index(part)
    # a full-size network would be 1.09 GB of tables for one part's test: build the one embedder
    from invr.network import Embedder
    e = Embedder(params.part_grid_spec(cfg, part), pid, part).to(dev)
    sd = _sd(part, full, None)
    with torch.no_grad():
        e.dense.copy_(sd['e.dense'].to(dev)); e.hash.copy_(sd['e.hash'].to(dev))
    x = torch.from_numpy(full['part_%s_x' % part]).to(dev)
    ref = full['part_%s_emb' % part]
    tol = _tolerance(ref, full['part_%s_entries_num' % part])
    generic = e(x).cpu().numpy()                                                       # invr_grid_encode_fwd
    assert (np.abs(generic[:, 3:] - ref[:, 3:]) <= tol).all() and np.abs(generic[:, :3] - ref[:, :3]).max() < 1e-6
    n = x.shape[0]
    L = _abi.lib()
    names = {0: 'k_part_encode_rs_xcd', 1: 'k_part_encode (64-byte rows)', 2: 'k_part_encode_rs'}
    for kernel in (0, 1, 2):
        keep = []
        g = e.grid_struct(keep)
        if kernel != 1:
            rs = e.row_sums()
            g.row_sums = rs.data_ptr()
        nbytes = L.invr_part_encode_workspace(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.full((n, 19), float('nan'), device=dev)
        _abi.check(L.invr_part_encode_fwd(C.byref(g), _abi.ptr(x), n, kernel, _abi.ptr(out), C.c_void_p(ws.data_ptr()), nbytes, _abi.stream_ptr()))
        got = out.cpu().numpy()
        rows = names[kernel]
        err = np.abs(got[:, 3:] - ref[:, 3:])
        assert np.abs(got[:, :3] - ref[:, :3]).max() < 1e-6
        assert (err <= tol).all(), (part, rows, float((err / tol).max()))
        inside = (tol <= 4.5e-6).all(1)
        assert inside.sum() > 3000 and err[inside].max() <= 4e-6, (part, rows, float(err[inside].max()))
