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


def test_renderer_bounds_the_memory_behind_unread_dicts():
    """Renderer._track_lazy: dicts whose raw / occ nobody read are completed on the host, oldest first, once the tensors they keep
    alive exceed lazy_device_budget (device_bytes counts CUDA storages; on the CPU the bookkeeping is exercised with a stand-in)."""
    from invr.renderer import Renderer

    class R(LazyHostRet):
        def device_bytes(self):
            return 0 if self._pending is not None else sum(v.numel() * 4 for k, v in self._lazy.items() if k == 'raw')
    r = Renderer.__new__(Renderer)
    r.lazy_device_budget, r._lazy_live = 2 * 8 * 4, []
    rets = []
    for i in range(5):
        host = {'rgb_map': torch.zeros(1, 2, 3)}
        raw = torch.full((1, 2, 4), float(i))
        rets.append(r._track_lazy(R(host, {'raw': raw, 'occ': raw[..., 3:]}, pin=False)))
    assert [x.pending() for x in rets[:3]] == [(), (), ()] and all(set(x.pending()) == {'raw', 'occ'} for x in rets[3:])
    assert all(float(x['raw'][0, 0, 0]) == i for i, x in enumerate(rets))            # values survive either way
    del rets
    r._track_lazy(R({}, {'raw': torch.zeros(1, 2, 4)}, pin=False))
    assert len(r._lazy_live) == 1                                                  # dead dicts are dropped from the book


def test_lazy_train_ret_dict_views_see_thunks_and_lazy_tensors():
    from invr.autograd import LazyTrainRet
    calls = []

    def mat():
        calls.append(1)
        return {'resd': torch.ones(1, 2, 3), 'tocc': torch.zeros(1, 2, 1)}
    mk = lambda: LazyTrainRet({'rgb_map': torch.zeros(1, 2, 3)}, ['resd', 'tocc'], mat, {'offset_loss': lambda: torch.tensor(2.0)})
    r = mk()
    assert len(r) == 4 and 'offset_loss' in r and 'resd' in r and not calls
    assert set(dict(r)) == {'rgb_map', 'resd', 'tocc', 'offset_loss'} and len(calls) == 1
    for use in (lambda x: {k: v for k, v in x.items()}, lambda x: dict(zip(x.keys(), x.values())), lambda x: {k: x[k] for k in x},
                lambda x: x.copy(), lambda x: copy.deepcopy(x), lambda x: pickle.loads(pickle.dumps(x))):
        d = use(mk())
        assert type(d) is dict and set(d) == {'rgb_map', 'resd', 'tocc', 'offset_loss'} and float(d['offset_loss']) == 2.0
    r = mk()
    assert float(r.pop('offset_loss')) == 2.0 and 'offset_loss' not in r and len(r) == 3


def test_lane_raw_buffer_is_reused_only_when_unreferenced():
    """Renderer lanes (in_flight > 1) hand the previous frame's raw buffer to the next frame only when no dict / tensor views it any
    more (storage use count) — a caller that kept `ret['raw']` keeps its data."""
    from invr.renderer import _Lane
    lane = _Lane.__new__(_Lane)
    lane.raw_buf = None
    dev = torch.device('cpu')
    a = lane.raw_buffer(1000, dev)
    assert a.numel() >= 1000 and lane.raw_buffer(900, dev) is a                     # nobody else references it: reused
    view = a[:400].view(100, 4)                                                    # (what a returned dict holds)
    b = lane.raw_buffer(900, dev)
    assert b is not a and b.untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
    del view, a
    assert lane.raw_buffer(900, dev) is b
    c = lane.raw_buffer(10 * b.numel(), dev)                                       # a larger frame: a larger buffer
    assert c is not b and c.numel() >= 10 * b.numel()


class _StubPending:
    """a frame still in flight (what Renderer.in_flight > 1 hands to LazyHostRet): result() joins it"""
    def __init__(self):
        self.joined = 0

    def done(self):
        return self.joined > 0

    def result(self):
        self.joined += 1
        raw = torch.arange(8.).reshape(1, 2, 4)
        return {'rgb_map': torch.arange(6.).reshape(1, 2, 3), 'acc_map': torch.ones(1, 2)}, {'raw': raw, 'occ': raw[..., 3:]}


def test_first_whole_dict_access_of_a_frame_in_flight_has_every_key():
    """keys() / items() / values() / iteration / dict(ret) / == / copy / pickling on a dict whose frame is still in flight
    must join the frame BEFORE deciding what to fetch — the first such access used to return rgb_map / acc_map only."""
    full = {'rgb_map', 'acc_map', 'raw', 'occ'}
    uses = (lambda r: dict(r), lambda r: dict(r.items()), lambda r: {k: r[k] for k in r}, lambda r: r.copy(), lambda r: dict(zip(r.keys(), r.values())),
            lambda r: copy.deepcopy(r), lambda r: pickle.loads(pickle.dumps(r)), lambda r: dict(zip(list(r), r.values())))
    for use in uses:
        p = _StubPending()
        r = LazyHostRet({}, {}, pin=False, pending=p, keys=sorted(full))
        assert len(r) == 4 and r.in_flight() and set(r.pending()) == full and p.joined == 0
        d = use(r)
        assert p.joined == 1 and type(d) is dict and set(d) == full, (set(d), full)
        assert all(not v.is_cuda for v in d.values()) and r.pending() == ()
    p = _StubPending()
    r = LazyHostRet({}, {}, pin=False, pending=p, keys=sorted(full))
    r2, _, _ = make()
    assert set(r.keys()) == full and (r == r) and set(r2.keys()) == full
    r = LazyHostRet({}, {}, pin=False, pending=_StubPending(), keys=sorted(full))
    r.fetch()
    assert r.pending() == () and set(dict.keys(r)) == full
