"""invr — MI355X-native per-ray render path for Instant-NVR."""
from . import config  # noqa: F401

__all__ = ['config']
