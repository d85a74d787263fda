"""The C-ABI library loads without a GPU and exports every function include/aq_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "aq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aq_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for n in ("aq_synth_fwd", "aq_synth_bwd", "aq_conv_fwd", "aq_conv_fwd_codes", "aq_conv_dgrad", "aq_conv_wgrad",
              "aq_linear_fwd", "aq_linear_bwd", "aq_last_error", "aq_abi_version"):
        assert n in names


def test_library_exports_every_declared_symbol():
    import quantized_cnn_amd._lib as L
    assert os.path.exists(L.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    dll = ctypes.CDLL(L.LIB_PATH)
    for n in _declared():
        assert hasattr(dll, n), "libaq_hip.so does not export %s" % n
    assert sorted(L.EXPORTS) == _declared()
    dll.aq_abi_version.restype = ctypes.c_int
    assert dll.aq_abi_version() == L.ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens before any launch, so these calls are safe on a CPU-only host."""
    import quantized_cnn_amd._lib as L
    lib = L.lib()
    segs = (L.AqSeg * 1)()
    segs[0].rows, segs[0].L = 4, 500
    assert lib.aq_synth_fwd(segs, 1, 1.0, 0, None, 0, None) < 0
    assert b"L=500" in lib.aq_last_error()
    segs[0].L = 32
    assert lib.aq_synth_fwd(segs, 1, 1000.0, 0, None, 0, None) < 0
    assert b"beta" in lib.aq_last_error()
    assert lib.aq_conv_fwd(None, None, None, 1, 3, 8, 8, 4, 4, 1, 2, None) < 0
    assert b"3x3" in lib.aq_last_error()
    assert lib.aq_conv_fwd(None, None, None, 1, 3, 8, 8, 4, 5, 1, 2, None) < 0  # a supported shape: fails on the pointers
    assert b"NULL" in lib.aq_last_error()
    assert lib.aq_conv_wgrad_workspace_bytes(128, 16, 32, 32, 16, 3, 1, 1) > 0


def test_cpu_tensors_are_refused_loudly():
    import pytest
    import torch
    import quantized_cnn_amd.ops as ops
    p, q = torch.rand(4, 32), torch.rand(32)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.synth_forward([p], [q], 1.0)
