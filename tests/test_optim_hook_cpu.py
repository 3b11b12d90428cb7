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
