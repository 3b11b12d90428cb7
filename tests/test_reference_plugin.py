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
