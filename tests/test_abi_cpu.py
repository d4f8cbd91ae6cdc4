"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/tnqs.h declares,
and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tnqs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(tnqs_[a-z0-9_]+)\s*\(", src))
    names.discard("tnqs_allgatherv_fn")
    return names


def test_library_exports_every_declared_symbol():
    import ctypes
    import tnqs_amd as tn
    lib = ctypes.CDLL(tn.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/tnqs.h but not exported"
    assert set(tn.EXPORTS) == decl
    assert lib.tnqs_version() >= 100


def test_no_cpu_fallback():
    import torch
    import tnqs_amd as tn
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = tn.named_grid((2, 2))
    psi = tn.tensornetworkstate(np.complex64, lambda v: "↑", g)
    with pytest.raises(tn.TnqsError, match="no HIP device"):
        tn.BeliefPropagationCache(psi)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "tensornetworkquantumsimulator.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "tnqs_oracle" not in txt and "import statevector" not in txt and "oracle/" not in txt, f


def test_c_driver_compiles_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/c_driver.c (plain C, gcc) links against libtnqs_hip.so through include/tnqs.h alone; without a GPU the library
    refuses to run (no CPU fallback) and the driver reports the library's error string"""
    import subprocess
    import torch
    exe = str(tmp_path / "c_driver")
    pkg = os.path.join(ROOT, "tensornetworkquantumsimulator.jl_amd")
    r = subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_driver.c"), "-o", exe,
                        "-L" + pkg, "-ltnqs_hip", "-lm", "-Wl,-rpath," + pkg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by tests/test_gpu_parity.py")
    p = subprocess.run([exe, "3", "2"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr
