"""CPU checks of the drop-in boundary: the shared library builds, loads without a GPU and exports
every symbol include/deeprl_amd.h declares; product code never imports the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from deeprl_amd import _lib
    if not os.path.isfile(_lib.LIBRARY):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "deeprl_amd", "csrc"), "-j8"])
    protos = _lib.parse_header()
    assert len(protos) >= 30
    dll = ctypes.CDLL(_lib.LIBRARY)
    for name in protos:
        assert hasattr(dll, name), "missing export %s" % name
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIBRARY]).decode()
    exported = set(re.findall(r" T (dra_\w+)", exported))
    assert exported == set(protos), "header and library disagree: %s" % sorted(exported ^ set(protos))


def test_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    from deeprl_amd._lib import DraError, lib
    h = ctypes.c_void_p()
    with pytest.raises(DraError):
        lib.dra_sumtree_create(ctypes.byref(h), 16)  # hipErrorNoDevice -> exception, never a CPU fallback


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "deeprl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ref_shim" in src:
                    bad.append(os.path.join(base, f))
    assert not bad, bad
