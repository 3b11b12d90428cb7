"""`cfg.renderer_module` target: exports `Renderer` (lib/networks/renderer/make_renderer.py:5-16)."""
from . import _config  # noqa: F401
from ..renderer import Renderer  # noqa: F401
