"""crabml_amd -- MI355X (gfx950) backend for crabml's block-quantized decode hot path.

The package is a thin python face over two in-tree native libraries:

  libcrabml_hip.so   the C-ABI tensor backend (include/crabml_hip.h): hand-written HIP kernels
  _host*.so          the host-side mirror of crabml's `Tensor` trait + `Llama2Runner` (C++/pybind11),
                     which calls the C ABI exactly as the `crabml-hip` Rust crate would

There is NO CPU fallback: importing works without a GPU (so the ABI can be inspected), but creating a
HipTensorDevice raises unless a HIP device is present, and a missing native library is an ImportError.
"""
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "libcrabml_hip.so")

if not _os.path.exists(LIB_PATH):
    raise ImportError(
        "crabml_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). The hip backend has no CPU fallback." % LIB_PATH)

from . import _host  # noqa: E402  (raises ImportError loudly if the extension was not built)
from ._host import (  # noqa: E402,F401
    CrabmlError, GGMLType, GGUFFile, HipLlamaRunner, HipTensor, HipTensorDevice, Llama2Runner, LlamaConfig, LlamaWeights, RopeMode,
    TensorStrider, TpComm, abi_version, sample_argmax,
)

__all__ = ["CrabmlError", "GGMLType", "GGUFFile", "HipLlamaRunner", "HipTensor", "HipTensorDevice", "Llama2Runner", "LlamaConfig", "LlamaWeights",
           "RopeMode", "TensorStrider", "TpComm", "abi_version", "sample_argmax", "LIB_PATH"]
