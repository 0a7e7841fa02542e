"""The C-ABI library loads without a GPU and exports exactly what include/vitsmi.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "vitsmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vits_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    names = _declared()
    assert "vits_mas_f32" in names
    handle = ctypes.CDLL(pkg._lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in vitsmi.h but not exported"


def test_binding_covers_header(pkg):
    assert sorted(pkg._lib.SIGNATURES) == _declared()
    assert pkg._lib.lib().vits_abi_version() == pkg._lib.ABI_VERSION


def test_rejects_bad_args_without_gpu(pkg):
    lib = pkg._lib.lib()
    assert lib.vits_mas_f32(None, None, 0, None, None, 1, 1, 1, None, None) == -1


def test_ops_fail_loudly_on_cpu_tensors(pkg):
    import pytest
    import torch
    with pytest.raises(RuntimeError, match="HIP kernels"):
        pkg.monotonic_align.maximum_path(torch.zeros(1, 2, 2), torch.ones(1, 2, 2))
