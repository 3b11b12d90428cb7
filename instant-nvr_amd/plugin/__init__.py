"""Modules the reference's factories import by yaml string:
    network_module invr.plugin.network    renderer_module invr.plugin.renderer
On import the host application's parsed config (lib.config.cfg) is adopted, if there is one."""
import sys

from .. import config as _config

_host = sys.modules.get('lib.config')
if _host is not None and hasattr(_host, 'cfg'):
    _config.adopt(_host.cfg)
