"""invr — MI355X-native per-ray render path for Instant-NVR."""
import os as _os

# ROCm 7.0's "graph packet capture" replay path (a graph without parallel branches is replayed from pre-built AQL packets) faults — a GPU
# memory access fault on an address of no allocation of ours — when replays of such a graph are interleaved with RCCL operations
# (profiles/r4_front_chain.md "runtime fault": tests/test_gpu_rccl_world1.py reproduced it with the single-stream frame of round 4 and
# passes with the switch off; replay times are the same either way).  The runtime reads the switch when HIP initialises, i.e. at the
# process's first device call: importing this package before that is enough.  An explicit setting in the environment wins.
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

from . import config  # noqa: F401,E402

__all__ = ['config']
