"""CPU: the zero-edit optimizer adoption (invr.optim.adopt_on_first_step) leaves every optimizer alone that is not a plain
torch.optim.Adam over exactly a registered network's DEVICE parameters — here: a CPU network, a partial parameter set, another class,
an unsupported Adam variant — and can be switched off."""
import os

import torch

from invr import optim
from invr.config import make_cfg
from invr.network import Network


def _steps(opt, params):
    for p in params:
        p.grad = torch.ones_like(p) * 1e-3
    opt.step()


def test_hook_ignores_what_it_must_not_touch(monkeypatch):
    net = Network(cfg=make_cfg(table_log2=6)).train()
    assert optim.adopt_on_first_step(net) and optim._HOOK[0] is not None and net in optim._ADOPT_NETS
    params = [p for p in net.parameters() if p.requires_grad]
    small = params[-4:]
    for build in (lambda: torch.optim.Adam([{'params': [p]} for p in params], 1e-3, eps=1e-15),             # CPU parameters: the fused step needs the device
                  lambda: torch.optim.Adam(small, 1e-3),                                                   # not the network's parameter set
                  lambda: torch.optim.SGD(small, 1e-3),
                  lambda: torch.optim.Adam(small, 1e-3, amsgrad=True)):
        opt = build()
        _steps(opt, small)
        assert getattr(opt, '_invr_inner', None) is None
        assert 'step' not in opt.__dict__                       # the instance's own step() was not rebound
    assert not optim._adam_matches(torch.optim.Adam(params, 1e-3), net)          # (CPU tensors)


def test_hook_can_be_switched_off(monkeypatch):
    monkeypatch.setenv('INVR_NO_OPTIM_HOOK', '1')
    net = Network(cfg=make_cfg(table_log2=6))
    assert optim.adopt_on_first_step(net) is False and net not in optim._ADOPT_NETS
    monkeypatch.delenv('INVR_NO_OPTIM_HOOK')
    cfg = make_cfg(table_log2=6)
    cfg['fused_optimizer_hook'] = False
    net2 = Network(cfg=cfg)
    assert optim.adopt_on_first_step(net2) is False


def test_unsupported_variants_and_closures_are_left_to_torch(monkeypatch):
    """decoupled_weight_decay (AdamW semantics on torch.optim.Adam, torch >= 2.7) must not be adopted or fused — the kernel
    applies L2 decay —, and a step(closure) is never the adopting step (torch runs the closure after the hook)."""
    import inspect
    net = Network(cfg=make_cfg(table_log2=6)).train()
    params = [p for p in net.parameters() if p.requires_grad]
    monkeypatch.setattr(optim, '_adam_matches', lambda opt, n, _orig=optim._adam_matches: _orig(opt, n))
    if 'decoupled_weight_decay' in inspect.signature(torch.optim.Adam.__init__).parameters:
        opt = torch.optim.Adam([{'params': [p]} for p in params], 1e-3, weight_decay=1e-2, decoupled_weight_decay=True)
        assert not optim._adam_matches(opt, net) and optim.fuse(opt, net) is opt
    # a group dict carrying the flag is refused whatever torch's constructor accepts
    opt = torch.optim.Adam([{'params': [p]} for p in params], 1e-3)
    opt.param_groups[3]['decoupled_weight_decay'] = True
    assert not optim._adam_matches(opt, net) and optim.fuse(opt, net) is opt
    # step(closure): the hook returns before looking at the optimizer at all
    optim.adopt_on_first_step(net)
    called = []
    monkeypatch.setattr(optim, 'adopt', lambda *a, **k: called.append(1))
    monkeypatch.setattr(optim, '_adam_matches', lambda *a: True)
    opt = torch.optim.Adam(params[-2:], 1e-3)
    # (torch's convention: args = the step wrapper's positional arguments, the optimizer itself first)
    assert optim._step_pre_hook(opt, (opt, lambda: None), {}) is None and optim._step_pre_hook(opt, (opt,), {'closure': lambda: None}) is None
    assert not called
    # ... and through the real hook path: a closure-free step() of a matching optimizer DOES reach adopt (the guard must not eat it)
    monkeypatch.setattr(optim, 'adopt', lambda o, n: (called.append(2), type('I', (), {'step': lambda self: None})())[1])
    for p_ in params[-2:]:
        p_.grad = torch.zeros_like(p_)
    opt.step()
    assert called == [2]
    called.clear()
    opt2 = torch.optim.Adam(params[-2:], 1e-3)
    opt2.step(lambda: 0.0)
    assert not called


def test_adoption_guard_reads_the_running_torch():
    """the private marks the adoption sets exist in this torch (else the guard says no, warns once and torch's own step is kept)"""
    assert optim._adoption_supported() is True
    import warnings
    from torch.optim import lr_scheduler
    src_has = hasattr(lr_scheduler, 'LRScheduler')
    assert src_has
    import inspect
    real = inspect.getsource
    try:
        inspect.getsource = lambda m: 'nothing of the kind'
        optim._WARNED[0] = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            assert optim._adoption_supported() is False
            assert any('adoption' in str(x.message) for x in w)
    finally:
        inspect.getsource = real
        optim._WARNED[0] = False
