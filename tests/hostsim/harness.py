"""TEST INFRASTRUCTURE — routes the Python host layer (invr._abi and everything above it) to the HOST build of the kernel sources
(tests/hostsim/build.py) for the duration of a `with activate():` block, with CPU tensors standing in for device memory.  The
product is not touched: the patches live in the test process only and are undone on exit; `invr._abi.ptr` keeps refusing CPU
tensors everywhere else."""
import contextlib
import ctypes as C

import torch


def _ptr(t, dtype=torch.float32):
    if t is None:
        return C.c_void_p(0)
    assert not t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, dtype)
    return C.c_void_p(t.data_ptr())


def _workspace(self, nbytes, device):
    """Network.workspace with the 256-byte alignment the library asks for (device allocations have it, host ones do not)."""
    if getattr(self, '_ws', None) is None or self._ws.numel() < nbytes or self._ws.is_cuda:
        raw = torch.empty(nbytes + 256, dtype=torch.uint8)
        off = (-raw.data_ptr()) % 256
        self._ws_raw, self._ws = raw, raw[off:off + nbytes]
    self._ws_gen = getattr(self, '_ws_gen', 0) + 1
    if _WS_FILL is not None:                               # HOSTSIM_WS_FILL=<byte>: every hand-out is junk (nothing may read what it has not written)
        self._ws.fill_(_WS_FILL)
    if _ASAN is not None:                                  # sanitizer runs: forget the red zones of the previous call's layout
        _ASAN.__asan_unpoison_memory_region(C.c_void_p(self._ws.data_ptr()), C.c_size_t(self._ws.numel()))
    return self._ws


import os as _os
_WS_FILL = int(_os.environ['HOSTSIM_WS_FILL'], 0) if _os.environ.get('HOSTSIM_WS_FILL') else None

try:
    _ASAN = C.CDLL(None)
    _ASAN.__asan_unpoison_memory_region
except (OSError, AttributeError):
    _ASAN = None


class Counters:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        for n in ('hostsim_anomalies', 'hostsim_split_waves', 'hostsim_launches'):
            getattr(self.lib, n).restype = C.c_long

    anomalies = property(lambda s: s.lib.hostsim_anomalies())       # reads of a lane that was not taking part in the operation
    split_waves = property(lambda s: s.lib.hostsim_split_waves())   # wave operations resolved while other lanes waited elsewhere
    launches = property(lambda s: s.lib.hostsim_launches())

    def reset(self):
        self.lib.hostsim_reset_counters()


@contextlib.contextmanager
def activate(extra_flags=()):
    """HOSTSIM_FLAGS in the environment adds compiler flags (tools/hostsim_asan.sh: the sanitizer builds)."""
    import os
    import platform
    from . import build
    if platform.machine() != 'x86_64' or not os.path.exists(build.CXX):
        import pytest
        pytest.skip('the wave machine needs an x86-64 host and the ROCm clang++ (tests/hostsim/hostsim_rt.cpp switches fibers in assembly)')
    extra_flags = tuple(extra_flags) + tuple(os.environ.get('HOSTSIM_FLAGS', '').split())
    from invr import _abi
    from invr.network import Network
    path = build.build(extra=tuple(extra_flags))
    from invr import driver
    make_opt = driver.make_optimizer
    saved = (_abi.LIB_PATH, _abi._lib, _abi.ptr, _abi.stream_ptr, Network.workspace, torch.cuda.synchronize, make_opt)
    _abi.LIB_PATH, _abi._lib = path, None
    _abi.ptr, _abi.stream_ptr = _ptr, (lambda: C.c_void_p(0))
    Network.workspace = _workspace
    torch.cuda.synchronize = lambda *a, **k: None
    # the driver picks the fused optimiser for device-resident parameters: here the "device" is the host build
    driver.make_optimizer = lambda net, *a, **k: make_opt(net, *a, **dict(k, fused=k.get('fused') if k.get('fused') is not None else True))
    try:
        _abi.lib()
        yield Counters(path)
    finally:
        _abi.LIB_PATH, _abi._lib, _abi.ptr, _abi.stream_ptr, Network.workspace, torch.cuda.synchronize, driver.make_optimizer = saved
