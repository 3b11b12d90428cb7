"""CPU: dict semantics of the eval return object (invr.renderer.LazyHostRet): the reference's callers treat the return of
Renderer.render as a plain dict of host tensors (evaluators/if_nerf.py:77, visualizers/if_nerf.py:24 index it; run.py iterates it)."""
import copy
import pickle

import torch

from invr.renderer import LazyHostRet


def make():
    host = {'rgb_map': torch.arange(6.).reshape(1, 2, 3), 'acc_map': torch.ones(1, 2)}
    lazy = {'raw': torch.arange(8.).reshape(1, 2, 4), 'occ': torch.arange(8.).reshape(1, 2, 4)[..., 3:]}      # (occ: a strided view, as the renderer hands it over)
    return LazyHostRet(host, lazy, pin=False), host, lazy


def test_access_by_key_fetches_only_that_key():
    r, host, lazy = make()
    assert set(r.pending()) == {'raw', 'occ'} and len(r) == 4 and 'raw' in r and 'nope' not in r
    assert r['rgb_map'] is host['rgb_map'] and set(r.pending()) == {'raw', 'occ'}
    assert torch.equal(r['occ'], lazy['occ']) and r['occ'].is_contiguous() and r.pending() == ('raw',)
    assert r.get('nope', 5) == 5 and torch.equal(r.get('raw'), lazy['raw']) and r.pending() == ()


def test_whole_dict_views_fetch_everything():
    for use in (lambda r: dict(r), lambda r: {k: v for k, v in r.items()}, lambda r: {k: r[k] for k in r}, lambda r: r.copy(),
                lambda r: dict(zip(r.keys(), r.values())), lambda r: copy.deepcopy(r), lambda r: pickle.loads(pickle.dumps(r))):
        r, host, lazy = make()
        d = use(r)
        assert type(d) is dict and set(d) == {'rgb_map', 'acc_map', 'raw', 'occ'}
        assert torch.equal(d['raw'], lazy['raw']) and torch.equal(d['occ'], lazy['occ'])


def test_pop_eq_repr():
    r, host, lazy = make()
    assert 'on the device' in repr(r)
    assert torch.equal(r.pop('raw'), lazy['raw']) and 'raw' not in r and len(r) == 3
    assert r.pop('nope', None) is None
    r2, _, _ = make()
    r2.pop('raw')
    assert set(r.keys()) == set(r2.keys())
