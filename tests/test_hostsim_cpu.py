"""CPU: the kernel SOURCES of instant-nvr_amd/csrc, compiled for the host and executed wave by wave on a fiber machine
(tests/hostsim/: ballot / shuffle / DPP / readlane / MFMA as rendezvous of the 64 lanes of a wave), run the parity tests of
tests/test_gpu_parity.py at the golden sizes — the same test bodies, against the same reference goldens and oracle.  What this
covers that the other CPU tests cannot: the logic of the kernels themselves (index arithmetic, list building, merge rule, wave
scans, MFMA operand layouts, LDS carving) on a box without a GPU.  What it does not cover: timing, the hardware's transcendental
pipes (libm here), real concurrency.  Test infrastructure only — the product never loads the host build."""
import os

import pytest
import torch

import tests.test_gpu_parity as T
import tests.test_gpu_training as TT
from tests.hostsim import harness
from tests.test_gpu_parity import gpu_setup  # noqa: F401  (fixture)

# every test of the parity module; of the training module the ones at the golden's / a toy size (the others need the 1.09 GB model)
# (test_lan_config_training_loop_vs_oracle passes here too: 14 optimiser steps against the oracle's autograd, 110 s — left to -m gpu)
BORROWED = [(T, None), (TT, ['test_pair_term_gradient_float64_arbitration', 'test_reference_step_form_with_disabled_grad_scaler',
                             'test_random_bg_epsilon_training_steps_vs_oracle', 'test_aggr_mean_training_steps_vs_oracle',
                             'test_aggr_distance_merges_training_steps_vs_oracle'])]


@pytest.fixture(scope='module', autouse=True)
def hostsim():
    old = [m.DEV for m, _ in BORROWED]
    for m, _ in BORROWED:
        m.DEV = 'cpu'
    try:
        with harness.activate() as counters:
            yield counters
            # no kernel read a lane that was not taking part in the operation (readlane / shuffle from a disabled lane)
            assert counters.anomalies == 0, counters.anomalies
    finally:
        for (m, _), d in zip(BORROWED, old):
            m.DEV = d


# The default CPU suite leaves out the slowest bodies (8-10 s each on the wave machine; HOSTSIM_FULL=1 runs everything — they pass)
SLOW = set() if os.environ.get('HOSTSIM_FULL') else {'test_network_wrapper_optimisation_steps', 'test_network_forward_on_points',
                                                     'test_reference_step_form_with_disabled_grad_scaler'}
for _m, _names in BORROWED:
    for _n in (_names or [n for n in dir(_m) if n.startswith('test_')]):
        if _n not in SLOW:
            globals()['test_hostsim__' + _n[5:]] = getattr(_m, _n)


def test_hostsim_results_do_not_depend_on_lane_or_wave_order():
    """The wave machine runs the lanes of a wave one after the other between rendezvous, and the waves of a workgroup one after the
    other between barriers.  On the hardware the lanes run in lockstep and the waves concurrently, so no result may depend on the
    order the machine happens to use: the golden render and the gradient goldens again with the lanes in a pseudo-random order
    and the waves reversed (a separate process: the order is fixed when the library loads).  This is what found the two places
    where lanes hand data to each other through LDS with no wave-level operation in between (k_encode.hip: wave barriers)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HOSTSIM_LANE_ORDER='shuffle:7', HOSTSIM_WAVE_ORDER='reverse')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider', '-k',
                        'render_64x64x32 or train_step_gradients or part_fields or knn_fallback'],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
