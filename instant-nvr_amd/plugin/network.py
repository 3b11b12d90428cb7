"""`cfg.network_module` target: exports `Network` (lib/networks/make_network.py:5-8)."""
from . import _config  # noqa: F401  (adopts the host cfg)
from ..network import Network  # noqa: F401
