"""Regression tests for defects found in review: they pin behaviour of the C ABI that the llama forward
never exercises but the Tensor-trait boundary allows."""
import threading

import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def test_activation_quant_cache_is_keyed_on_the_row_length(ca, odev):
    """One unmodified f32 buffer used as rhs first as (b = 2, k = 64), then -- through a reshape view -- as (b = 1,
    k = 128): same byte count, same rhs type, but the plane layout depends on k.  The cache used to hit."""
    hdev = ca.HipTensorDevice(0, False, 0, True)  # strict: both results are compared bit for bit
    rng = np.random.default_rng(5)
    x = rng.standard_normal(128).astype(np.float32)
    w64 = synth.random_blocks(rng, 8 * 64, synth.Q8_0)
    w128 = synth.random_blocks(rng, 8 * 128, synth.Q8_0)
    hx = ca.HipTensor.new(x, [2, 64], hdev)
    a = ca.HipTensor.from_cpu(w64, [8, 64], ca.GGMLType.Q8_0, hdev).matmul_vec(hx).export()
    b = ca.HipTensor.from_cpu(w128, [8, 128], ca.GGMLType.Q8_0, hdev).matmul_vec(hx.reshape([1, 128])).export()
    ra = o.OracleTensor.from_bytes(w64, o.Q8_0, [8, 64], odev).matmul_vec(o.OracleTensor.new(x, [2, 64], odev)).export()
    rb = o.OracleTensor.from_bytes(w128, o.Q8_0, [8, 128], odev).matmul_vec(o.OracleTensor.new(x, [1, 128], odev)).export()
    assert np.array_equal(a.view(np.uint32), ra.view(np.uint32))
    assert np.array_equal(b.view(np.uint32), rb.view(np.uint32))


def test_entry_points_select_their_device_when_driven_from_another_thread(ca, odev):
    """A fresh host thread has device 0 current whatever the creating thread selected; every entry point that allocates
    or launches therefore selects the HipTensorDevice's own ordinal (CH_USE).  With one GPU the ordinal is 0 either way,
    so this checks the thread path end to end; on a multi-GPU box it runs on the LAST device."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
    ordinal = n.value - 1
    dev = ca.HipTensorDevice(ordinal, False, 0, True)
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=2)
    from tests.helpers import to_oracle

    oconf, ow = to_oracle(model, odev)
    ref = o.OracleLlamaRunner(oconf, ow, odev, 16, True).forward([3], 0).copy()
    out = {}

    def work():
        try:
            conf, w = synth.to_hip(model, dev)
            out["trait"] = np.asarray(ca.Llama2Runner(conf, w, dev, 16, True).forward([3], 0)).copy()
            out["fused"] = ca.HipLlamaRunner(conf, w, dev, 16, True).forward(3, 0).copy()
        except Exception as e:  # surfaced in the main thread
            out["error"] = e

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "error" not in out, out.get("error")
    assert np.array_equal(out["trait"].view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(out["fused"].view(np.uint32), ref.view(np.uint32))
