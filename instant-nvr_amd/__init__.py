"""invr — MI355X-native per-ray render path for Instant-NVR."""
import os as _os

# ROCm 7.0's "graph packet capture" replay path (a graph without parallel branches is replayed from pre-built AQL packets) faults — a GPU
# memory access fault on an address of no allocation of ours — when replays of such a graph are interleaved with RCCL operations
# (profiles/r4_front_chain.md "runtime fault": tests/test_gpu_rccl_world1.py reproduced it with the single-stream frame of round 4 and
# passes with the switch off; replay times are the same either way).  The runtime reads the switch when HIP initialises, i.e. at the
# process's first device call: importing this package before that is enough.  An explicit setting in the environment wins.
# A host that touched the GPU BEFORE importing this package (torch.cuda.init(), model.to('cuda'), DDP set-up, as train_net.py / run.py
# do) initialised HIP without the switch: PACKET_CAPTURE_OFF is False then, a warning says so, and invr.frames.FrameSet never builds a
# branch-free graph around a collective (it adds an empty side branch: one cross-stream edge, ~15 us per replay).
import sys as _sys
import warnings as _warnings


def _hip_initialised():
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


_late = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE' not in _os.environ and _hip_initialised()
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
PACKET_CAPTURE_OFF = _os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] == '0' and not _late
if _late:
    _warnings.warn('invr was imported after the first GPU call of this process: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 could not take effect '
                   '(export it, or import invr before touching the GPU); hipGraph replays around RCCL operations get a side branch instead '
                   '(INTEGRATION.md "import order")')

from . import config  # noqa: F401,E402

__all__ = ['config']
