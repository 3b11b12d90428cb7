"""`cfg.trainer_module` target: exports `NetworkWrapper` (lib/train/trainers/make_trainer.py:4-7)."""
from . import _config  # noqa: F401
from ..trainer import NetworkWrapper  # noqa: F401
