"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/crabml_hip.h (the drop-in surface) and include/crabml_hip_debug.h (parity / measurement hooks, A/B switches: test
infrastructure) declare; the public header stays thin; the host mirror imports; and without a GPU the backend fails loudly
(there is no CPU fallback to hide behind)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(headers=("crabml_hip.h", "crabml_hip_debug.h")):
    names = []
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(crabml_hip_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_public_header_is_a_thin_boundary():
    """include/crabml_hip.h = the trait surface + the decode step + the TP calls: no debug_* / prof_* entry point, and at most
    eight flag bits (round-3 verdict, item 5); everything a test or a lab needs beyond that lives in crabml_hip_debug.h."""
    public = declared_functions(("crabml_hip.h",))
    assert not [n for n in public if "_debug_" in n or "_prof_" in n]
    src = open(os.path.join(ROOT, "include", "crabml_hip.h")).read()
    flags = re.findall(r"^#define (CRABML_HIP_(?:LLAMA|FLAG)_[A-Z0-9_]+) ", src, flags=re.M)
    assert 1 <= len(flags) <= 8, flags
    dbg = declared_functions(("crabml_hip_debug.h",))
    assert dbg and all("_debug_" in n or "_prof_" in n for n in dbg), dbg


def test_header_declares_the_trait_surface():
    names = declared_functions()
    # one entry point per `Tensor` trait method that touches data (api.rs:11-79)
    for need in ["buf_from_cpu", "buf_alloc", "contiguous", "concatenate", "copy_rows_from", "export", "dup",
                 "rope_inplace", "rms_norm_inplace", "softmax_inplace", "silu_inplace", "gelu_inplace", "mul_inplace",
                 "add_inplace", "scale_inplace", "matmul_vec", "batch_matmul", "device_create", "device_destroy",
                 "last_error"]:
        assert "crabml_hip_" + need in names


def test_library_exports_every_declared_symbol():
    import crabml_amd
    lib = ctypes.CDLL(crabml_amd.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/crabml_hip.h but not exported: {missing}"
    lib.crabml_hip_abi_version.restype = ctypes.c_int
    assert lib.crabml_hip_abi_version() == 2


def test_no_extra_undeclared_exports():
    import subprocess
    import crabml_amd
    out = subprocess.run(["nm", "-D", "--defined-only", crabml_amd.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (crabml_hip_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == declared_functions()


def test_host_mirror_imports_and_strider_works_without_gpu():
    import crabml_amd as ca
    s = ca.TensorStrider([3, 4])
    assert s.strides() == [4, 1] and s.at([1, 0]) == 4
    with pytest.raises(ca.CrabmlError):
        s.reshape([4, 2])
    st = ca.TensorStrider([2, 3]).transpose([1, 0])
    assert st.shape() == [3, 2] and st.iter() == [0, 3, 1, 4, 2, 5] and not st.is_contiguous()
    s3 = ca.TensorStrider([3, 8, 3200]).resize([3, 0, 3200])
    assert s3.shape() == [3, 0, 3200] and s3.strides() == [3200 * 8, 3200, 1]


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (CPU) code."""
    pkg = os.path.join(ROOT, "crabml_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "crabml_oracle.h" in txt or "liboracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_device_creation_fails_loudly_without_gpu():
    import crabml_amd as ca
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(ca.CrabmlError):
        ca.HipTensorDevice()
