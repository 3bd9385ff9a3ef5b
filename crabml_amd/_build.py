"""Build recipes for the native pieces (all in-tree, so the built .so files travel to the GPU box).

  libcrabml_hip.so   hipcc --offload-arch=gfx950   crabml_amd/csrc/*.hip       the C-ABI backend
  _host*.so          g++ + pybind11                crabml_amd/csrc/host/*.cpp  host-side mirror of the
                                                                                reference's Tensor trait +
                                                                                Llama2Runner (calls the C ABI)
hipcc cross-compiles gfx950 without a GPU.  Objects are rebuilt only when a source/header is newer.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LIB = os.path.join(HERE, "libcrabml_hip.so")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
             "-Wno-unused-function", "-Wno-unused-result"]
# per-file: the MFMA GEMM keeps its accumulators in VGPRs (gfx950's register file is unified; the AGPR form costs a
# v_accvgpr_read/write per accumulator register and block: 32 of 137 VALU instructions per k-block)
EXTRA_FLAGS = {"gemm_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stdout + r.stderr


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + HIP_FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
    if force or jobs or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def host_ext_path() -> str:
    return os.path.join(HERE, "_host" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_host(force: bool = False) -> str:
    import pybind11

    out = host_ext_path()
    srcs = sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "host", "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not (force or _newer(out, srcs + hdrs + [LIB])):
        return out
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include")]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-Wall"] + inc + \
        srcs + ["-o", out, "-L" + HERE, "-lcrabml_hip", "-Wl,-rpath,$ORIGIN"]
    _run(cmd)
    return out


def build_all(force: bool = False, verbose: bool = False):
    build_hip(force, verbose)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", LIB, host_ext_path())
