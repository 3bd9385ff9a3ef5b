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
           "RopeMode", "TensorStrider", "TpComm", "abi_version", "sample_argmax", "pin_host_to_device_node", "LIB_PATH"]


def pin_host_to_device_node(device):
    """Runs this process's threads on the host NUMA node the GPU is attached to (what `numactl --cpunodebind` does) and says what
    it did.  The fused entry points do not care where the host sits; a host that drives the device token by token -- the
    reference's runner: ~840 recorded calls, 160 launches, a completion flag and 513 KB of logits per token -- is 5-10 % faster
    from the near socket (tools/trait_var.py)."""
    node = device.numa_node()
    if node < 0:
        return "unchanged (the device's NUMA node is unknown)"
    cpus = set()
    with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
        for part in f.read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= _os.sched_getaffinity(0)
    if not cpus:
        return "unchanged (no allowed cpu on NUMA node %d)" % node
    _os.sched_setaffinity(0, cpus)
    return "NUMA node %d of the GPU (%d cpus)" % (node, len(cpus))
