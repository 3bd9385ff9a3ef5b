"""Build hygiene of the HIP objects (CPU test, reads the compiled code objects): no kernel of the library may use scratch
(private segment) memory or spill registers.  A kernel that takes the address of a by-value argument struct, or indexes a
register array dynamically, silently gets a scratch frame -- which on MI355X cost the wo / ffn_down kernels +4 us per launch
when it happened (round 2: 664 -> 554 tokens/s) without failing a single parity test."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="needs the ROCm llvm tools")
def test_no_kernel_uses_scratch_or_spills(tmp_path):
    objs = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.o")))
    if not objs:
        pytest.skip("objects not built (run __graft_entry__.build())")
    bad, n_kernels = [], 0
    for o in objs:
        local = tmp_path / os.path.basename(o)
        shutil.copy(o, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(local)], capture_output=True, cwd=tmp_path)
        for co in glob.glob(str(local) + ".*amdgcn*"):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, flags=re.S):
                n_kernels += 1
                if int(m.group(2)) != 0 or int(m.group(3)) != 0:
                    bad.append((os.path.basename(o), m.group(1), int(m.group(2)), int(m.group(3))))
    assert n_kernels > 100, n_kernels
    assert not bad, bad
