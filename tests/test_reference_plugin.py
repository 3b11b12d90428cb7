"""Drop-in seam against the REAL reference factories (build container only: skipped where
/root/reference is absent, e.g. on the GPU box).  Runs in a subprocess because importing the
reference parses sys.argv and needs stand-ins for packages this image lacks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys, os
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests', 'golden'))
import make_golden as mg
rcfg = mg.import_reference(16, ['network_module', 'invr.plugin.network', 'renderer_module', 'invr.plugin.renderer'])
import torch
from lib.networks.make_network import make_network
from lib.networks.renderer.make_renderer import make_renderer
net = make_network(rcfg)                       # -> invr.plugin.network.Network()
import invr
from invr.network import Network
from invr.renderer import Renderer
assert type(net) is Network, type(net)
assert invr.config.cfg.N_samples == 16 and invr.config.cfg.smpl_thresh == rcfg.smpl_thresh
r = make_renderer(rcfg, net)
assert type(r) is Renderer and r.net is net
# checkpoint interchange with the reference's own network class
rcfg.network_module = 'lib.networks.bw_deform.inb_part_network_multiassign'
ref_net = make_network(rcfg)
sd = ref_net.state_dict()
assert list(sd.keys()) == list(net.state_dict().keys())
net.load_state_dict(sd, strict=True)
ref_net.load_state_dict(net.state_dict(), strict=True)
print('PLUGIN_OK', len(sd))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
def test_reference_factories_pick_up_invr_plugin():
    env = dict(os.environ, PYTHONBREAKPOINT='0')
    r = subprocess.run([sys.executable, '-c', SCRIPT % {'root': ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert 'PLUGIN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


CKPT_SCRIPT = r'''
import sys, os, tempfile
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests', 'golden'))
import make_golden as mg
rcfg = mg.import_reference(16)
import torch
from lib.networks.make_network import make_network
from lib.utils import net_utils
import invr
from invr import driver
from invr.network import Network
from invr.config import make_cfg
ref_net = make_network(rcfg)
d = tempfile.mkdtemp()
# reference -> ours: a checkpoint written by the reference's save_model loads into the drop-in network
opt = torch.optim.Adam(ref_net.parameters()); sch = torch.optim.lr_scheduler.ExponentialLR(opt, 0.9)
class Rec:
    def state_dict(self): return {'step': 3}
    def load_state_dict(self, s): self.s = s
net_utils.save_model(ref_net, opt, sch, Rec(), d, 4, last=True)
mine = Network(cfg=make_cfg(table_log2=mg.TABLE_LOG2, N_samples=16))
assert driver.load_network(mine, d) == 5
for (k, a), (k2, b) in zip(ref_net.state_dict().items(), mine.state_dict().items()):
    assert k == k2 and torch.equal(a, b), k
# ours -> reference: driver.save_model writes what the reference's load_network / load_model read
d2 = tempfile.mkdtemp()
with torch.no_grad():
    for p in mine.parameters():
        if p.dtype.is_floating_point: p.add_(0.25)
driver.save_model(mine, opt, sch, Rec(), d2, 7, last=False)
assert net_utils.load_network(ref_net, d2) == 8
for (k, a), (k2, b) in zip(ref_net.state_dict().items(), mine.state_dict().items()):
    assert torch.equal(a, b), k
assert net_utils.load_model(ref_net, opt, sch, Rec(), d2) == 8
print('CKPT_OK')
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')
def test_checkpoints_interchange_with_reference_io():
    env = dict(os.environ, PYTHONBREAKPOINT='0')
    r = subprocess.run([sys.executable, '-c', CKPT_SCRIPT % {'root': ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert 'CKPT_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
