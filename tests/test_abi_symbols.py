"""CPU: the C-ABI shared library loads and exports every symbol include/invr.h declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'invr.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(invr_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_header_symbols():
    from invr import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    L = _abi.lib()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_abi.EXPORTS) == names
    assert L.invr_version() == _abi.ABI_VERSION == 2
    import re
    hdr = open(os.path.join(ROOT, 'include', 'invr.h')).read()
    assert int(re.search(r'#define INVR_ABI_VERSION (\d+)', hdr).group(1)) == _abi.ABI_VERSION


def test_workspace_query_and_error_path():
    from invr import _abi
    L = _abi.lib()
    small = L.invr_workspace_bytes(4096, 64, 0)
    big = L.invr_workspace_bytes(8192, 64, 0)
    capped = L.invr_workspace_bytes(8192, 64, 1000)
    assert 0 < small < big and capped < big
    # argument errors are reported through the status code + invr_last_error (no exceptions, no GPU touched)
    import ctypes as C
    st = L.invr_render_fwd(None, None, None, None, None, None, None, 1, 8, None, None, None, None, None, None, None,
                           None, 0, 0, None)
    assert st != 0 and b'null' in L.invr_last_error()
    st = L.invr_composite_fwd(None, 4, 0, None, None, None, None)
    assert st != 0


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (checked against sizes computed from the header's field lists)."""
    import ctypes as C
    from invr import _abi
    assert C.sizeof(_abi.InvrGrid) == 3 * 8 + 4 * 4 + 8 + 16 * 4 + 16 * 4 + 16 * 8 + 3 * 4 + 4 + 8
    assert C.sizeof(_abi.InvrMlp) == 8 * 8 + 5 * 4 + 4
    assert C.sizeof(_abi.InvrPart) == C.sizeof(_abi.InvrGrid) + 2 * C.sizeof(_abi.InvrMlp) + 8 + 8
    L = _abi.lib()
    assert C.sizeof(_abi.InvrAdamTensor) == 4 * 8 + 8 + 4 * 4 + 8 + 2 * 4
    assert C.sizeof(_abi.InvrTrainGrads) == 5 * (8 + 4 * 4 * 8 + 8) + 2 * 8 + 2 * 4 * 8 + 8
    for i, t in enumerate((_abi.InvrGrid, _abi.InvrMlp, _abi.InvrPart, _abi.InvrModel, _abi.InvrScene, _abi.InvrWsLayout, _abi.InvrMlpBwdOut,
                           _abi.InvrAdamTensor, _abi.InvrTrainGrads)):
        assert L.invr_sizeof(i) == C.sizeof(t), t


def test_product_path_has_no_cpu_route():
    """The render path is the HIP library or nothing: the binding refuses host tensors, and nothing in the product tree, bench.py or
    __graft_entry__.py knows about the test-only host build of the kernels (tests/hostsim) or imports the oracle outside bench.py's
    cpu_baseline leg / smoke()'s check."""
    import os
    import re
    import pytest
    import torch
    from invr import _abi
    with pytest.raises(AssertionError):
        _abi.ptr(torch.zeros(4))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')]
    for d, _, names in os.walk(os.path.join(root, 'instant-nvr_amd')):
        files += [os.path.join(d, n) for n in names if n.endswith(('.py', '.hip', '.h'))]
    for f in files:
        text = open(f).read()
        assert 'hostsim' not in text.lower(), f
        if not f.endswith(('bench.py', '__graft_entry__.py')):
            assert not re.search(r'^\s*(from|import)\s+(oracle|tests)\b', text, re.M), f
