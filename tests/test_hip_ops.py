"""GPU parity: every `Tensor` trait op of the hip backend against the CPU oracle, written like the
reference's own backend tests (crabml-wgpu/src/wgpu_tensor.rs:742-1099, cpu_tensor.rs:455-606), plus
seeded random parity.  All calls go HipTensor -> C ABI -> HIP kernels.  Bit-exact unless stated."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


def T(ca, hdev, v, shape):
    return ca.HipTensor.new(np.asarray(v, dtype=np.float32), shape, hdev)


def OT(odev, v, shape):
    return o.OracleTensor.new(np.asarray(v, dtype=np.float32), shape, odev)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------ reference goldens
def test_new_and_export(ca, hdev):  # wgpu_tensor.rs:761-771
    t = T(ca, hdev, [1, 2, 3, 4, 5, 6], [2, 3])
    assert t.export().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]


def test_add(ca, hdev):  # wgpu_tensor.rs:773-785
    t1 = T(ca, hdev, [2.0] * 64, [16, 4]).add_inplace(T(ca, hdev, [3.0] * 64, [16, 4]))
    assert t1.export().tolist() == [5.0] * 64


def test_mul(ca, hdev):  # wgpu_tensor.rs:787-808
    t1 = T(ca, hdev, [3.0] * 1024, [512, 2]).mul_inplace(T(ca, hdev, [2.0] * 1024, [512, 2]))
    assert np.all(t1.export() == 6.0)
    t1 = T(ca, hdev, [3.0] * 6, [3, 2]).mul_inplace(T(ca, hdev, [2.0] * 6, [3, 2]))
    assert t1.export().tolist() == [6.0] * 6


def test_alloc_is_zeroed(ca, hdev):  # wgpu_tensor.rs:810-821
    t1 = ca.HipTensor.alloc([512, 2], ca.GGMLType.F32, hdev).add_inplace(T(ca, hdev, [1.0] * 1024, [512, 2]))
    assert np.all(t1.export() == 1.0)
    with pytest.raises(ca.CrabmlError):  # cpu_tensor.rs:139-141
        ca.HipTensor.alloc([4], ca.GGMLType.Q8_0, hdev)


def test_with_name_debug_snapshot(ca):  # wgpu_tensor.rs:823-833
    dev = ca.HipTensorDevice(0, True)
    t1 = T(ca, dev, [0.0] * 1024, [512, 2]).add_inplace(T(ca, dev, [1.0] * 1024, [512, 2]))
    t1.with_name("t1")
    assert dev.dump_debug_tensor("t1").tolist() == [1.0] * 1024
    assert dev.dump_debug_tensor("nope") is None


def test_copy_rows_from(ca, hdev):  # wgpu_tensor.rs:835-851, cpu_tensor.rs:472-488
    t1 = ca.HipTensor.alloc([256, 4], ca.GGMLType.F32, hdev)
    t2 = T(ca, hdev, np.arange(1024), [256, 4])
    t1.copy_rows_from(t2, [1])
    assert t1.export()[0:4].tolist() == [4.0, 5.0, 6.0, 7.0]
    a = T(ca, hdev, [1, 2, 3, 4], [2, 2])
    b = T(ca, hdev, [0, 0], [2])
    b.copy_rows_from(a, [1])
    assert b.export().tolist() == [3.0, 4.0]
    b.copy_rows_from(a, [0])
    assert b.export().tolist() == [1.0, 2.0]


def test_rms_norm(ca, hdev, odev):  # wgpu_tensor.rs:853-877 (eps 1e-7 there); here: bit-exact vs oracle
    v = np.arange(1, 129, dtype=np.float32)
    got = T(ca, hdev, v, [128]).rms_norm_inplace(1e-5).export()
    ref = OT(odev, v, [128]).rms_norm_inplace(1e-5).export()
    assert np.array_equal(bits(got), bits(ref))
    with pytest.raises(ca.CrabmlError):  # rms_norm.rs:34 assert!(len % 32 == 0)
        T(ca, hdev, np.ones(48), [48]).rms_norm_inplace(1e-5)


def test_matmul_goldens(ca, hdev):  # wgpu_tensor.rs:880-894, cpu_tensor.rs:530-541
    t3 = T(ca, hdev, np.arange(256), [32, 8]).matmul_vec(T(ca, hdev, [2.0] * 8, [8]))
    assert t3.shape() == [32]
    assert t3.export().tolist() == [56.0 + 128.0 * i for i in range(32)]
    out = T(ca, hdev, [4.0] * 32, [16, 2]).matmul_vec(T(ca, hdev, [1.0, 2.0], [2]))
    assert out.export().tolist() == [12.0] * 16
    # (m,k) @ (b,k) -> (b,m)   (cpu_tensor.rs:374-378)
    out = T(ca, hdev, np.arange(256), [32, 8]).matmul_vec(T(ca, hdev, [2.0] * 8 + [1.0] * 8, [2, 8]))
    assert out.shape() == [2, 32]
    assert out.export().tolist() == [56.0 + 128.0 * i for i in range(32)] + [28.0 + 64.0 * i for i in range(32)]


def test_batch_matmul_golden(ca, hdev):  # wgpu_tensor.rs:897-915
    t1 = T(ca, hdev, np.arange(6), [1, 3, 2])
    t3 = t1.batch_matmul(T(ca, hdev, [2.0, 2.0], [1, 2, 1]))
    assert t1.strider().strides() == [6, 2, 1]
    assert t3.export().tolist() == [2.0, 10.0, 18.0]


def test_rope_golden(ca, hdev, odev):  # wgpu_tensor.rs:918-937, cpu_tensor.rs:509-527
    v = np.arange(32, dtype=np.float32)
    got = T(ca, hdev, v, [2, 16]).rope_inplace(ca.RopeMode.Llama, 1, 2).export()
    exp = np.array([-0.841471, 0.54030234] + list(range(2, 16)) + [-5.6601696, 22.648676] + list(range(18, 32)),
                   dtype=np.float32)
    assert np.allclose(got, exp, rtol=0, atol=1e-5)
    ref = OT(odev, v, [2, 16]).rope_inplace(o.ROPE_LLAMA, 1, 2).export()
    assert np.array_equal(bits(got), bits(ref))  # host libm cos/sin + same f32 ops -> bit-exact


def test_concatenate_goldens(ca, hdev):  # wgpu_tensor.rs:940-998
    t1 = ca.HipTensor.alloc([2, 2, 16], ca.GGMLType.F32, hdev).resize(0, 0)
    t1.concatenate(T(ca, hdev, np.arange(32), [1, 2, 16]), 0)
    t1.concatenate(T(ca, hdev, np.arange(32, 64), [1, 2, 16]), 0)
    assert t1.shape() == [2, 2, 16]
    assert t1.export().tolist() == [float(i) for i in range(64)]
    t1 = ca.HipTensor.alloc([2, 2, 16], ca.GGMLType.F32, hdev).resize(1, 0)
    t1.concatenate(T(ca, hdev, np.arange(32), [2, 1, 16]), 1)
    t1.concatenate(T(ca, hdev, np.arange(32, 64), [2, 1, 16]), 1)
    exp = list(range(0, 16)) + list(range(32, 48)) + list(range(16, 32)) + list(range(48, 64))
    assert t1.shape() == [2, 2, 16]
    assert t1.export().tolist() == [float(i) for i in exp]
    with pytest.raises(ca.CrabmlError):  # full cache
        t1.concatenate(T(ca, hdev, np.arange(32), [2, 1, 16]), 1)
    with pytest.raises(ca.CrabmlError):  # shape mismatch (cpu_tensor.rs:272-283)
        ca.HipTensor.alloc([2, 2, 16], ca.GGMLType.F32, hdev).resize(1, 0).concatenate(T(ca, hdev, np.arange(24), [2, 1, 12]), 1)


def test_softmax_golden(ca, hdev, odev):  # wgpu_tensor.rs:1000-1018 / cpu_tensor.rs:544-555
    v = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    got = T(ca, hdev, v, [2, 3]).softmax_inplace(1).export()
    assert np.allclose(got, [0.09003057, 0.24472848, 0.66524094] * 2, atol=1e-3, rtol=0)
    ref = OT(odev, v, [2, 3]).softmax_inplace(1).export()
    assert np.array_equal(bits(got), bits(ref))
    with pytest.raises(ca.CrabmlError):  # softmax.rs:21-28
        T(ca, hdev, v, [2, 3]).softmax_inplace(0)


def test_silu_gelu_goldens(ca, hdev, odev):  # wgpu_tensor.rs:1020-1056 / cpu_tensor.rs:558-569
    v = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    got = T(ca, hdev, v, [6]).silu_inplace().export()
    assert np.allclose(got, [0.7310586, 1.761594, 2.8577225, 3.928055, 4.9665356, 5.9851646], atol=1e-1, rtol=0)
    assert np.array_equal(bits(got), bits(OT(odev, v, [6]).silu_inplace().export()))
    got = T(ca, hdev, v, [6]).gelu_inplace().export()
    assert np.array_equal(bits(got), bits(OT(odev, v, [6]).gelu_inplace().export()))


def test_dup(ca, hdev):  # wgpu_tensor.rs:1058-1073
    t1 = T(ca, hdev, [1, 2, 3, 4, 5, 6], [2, 3])
    t2 = t1.dup()
    t1.scale_inplace(2.0)
    assert t2.export().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    assert t1.export().tolist() == [2.0, 4.0, 6.0, 8.0, 10.0, 12.0]


def test_contiguous(ca, hdev):  # wgpu_tensor.rs:1075-1098, cpu_tensor.rs:572-600
    t1 = T(ca, hdev, [1, 2, 3, 4, 5, 6], [2, 3]).transpose([1, 0])
    t2 = t1.contiguous()
    assert t2.strider().shape() == [3, 2] and t2.strider().dims() == 2
    assert t2.export().tolist() == [1.0, 4.0, 2.0, 5.0, 3.0, 6.0]
    t1 = T(ca, hdev, [1, 2, 3, 4, 5, 6], [1, 2, 3]).transpose([2, 1, 0])
    t2 = t1.contiguous()
    assert t2.export().tolist() == [1.0, 4.0, 2.0, 5.0, 3.0, 6.0] and t2.shape() == [3, 2, 1]
    with pytest.raises(ca.CrabmlError):
        t1.export()  # export asserts contiguity (cpu_tensor.rs:341)


def test_tensor_view_and_reshape_errors(ca, hdev):  # cpu_tensor.rs:461-470, strider.rs:143-160
    t = T(ca, hdev, [1, 2, 3, 4, 5, 6], [2, 3]).reshape([3, 2])
    assert t.reshape([2, 3]).export().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    with pytest.raises(ca.CrabmlError):
        t.reshape([4, 2])
    with pytest.raises(ca.CrabmlError):
        t.transpose([1, 0]).reshape([6])
    with pytest.raises(ca.CrabmlError):
        ca.HipTensor.new(np.zeros(5, dtype=np.float32), [2, 3], hdev)  # cpu_tensor.rs:31-38
    with pytest.raises(ca.CrabmlError):
        t.resize(5, 1)
    with pytest.raises(ca.CrabmlError):
        t.resize(0, 100)


# ------------------------------------------------------------------ seeded random parity (bit-exact)
@pytest.mark.parametrize("rows,cols", [(1, 288), (1, 4096), (3, 64), (2, 8192)])
def test_rms_norm_random_bitexact(ca, hdev, odev, rows, cols):
    rng = np.random.default_rng(rows * 100003 + cols)
    v = (rng.standard_normal(rows * cols) * 3).astype(np.float32)
    got = T(ca, hdev, v, [rows, cols]).rms_norm_inplace(1e-5).export()
    ref = OT(odev, v, [rows, cols]).rms_norm_inplace(1e-5).export()
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("mode", ["Llama", "Neox"])
@pytest.mark.parametrize("shape,pos,rope_dims", [((6, 48), 0, 48), ((6, 48), 17, 48), ((32, 128), 1023, 128),
                                                 ((8, 128), 5, 64), ((2, 4, 16), 3, 16), ((1, 8, 8), 100, 8)])
def test_rope_random_bitexact(ca, hdev, odev, mode, shape, pos, rope_dims):
    rng = np.random.default_rng(pos + 7)
    n = int(np.prod(shape))
    v = rng.standard_normal(n).astype(np.float32)
    hm = getattr(ca.RopeMode, mode)
    om = o.ROPE_LLAMA if mode == "Llama" else o.ROPE_NEOX
    got = T(ca, hdev, v, list(shape)).rope_inplace(hm, pos, rope_dims).export()
    ref = OT(odev, v, list(shape)).rope_inplace(om, pos, rope_dims).export()
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("shape", [(6, 1, 1), (6, 1, 37), (32, 1, 128), (32, 1, 1025), (4, 7)])
def test_softmax_random_bitexact(ca, hdev, odev, shape):
    rng = np.random.default_rng(sum(shape))
    v = (rng.standard_normal(int(np.prod(shape))) * 4).astype(np.float32)
    axis = len(shape) - 1
    got = T(ca, hdev, v, list(shape)).softmax_inplace(axis).export()
    ref = OT(odev, v, list(shape)).softmax_inplace(axis).export()
    assert np.array_equal(bits(got), bits(ref))


def test_elementwise_random_bitexact(ca, hdev, odev):
    rng = np.random.default_rng(11)
    a = (rng.standard_normal(4 * 768) * 5).astype(np.float32)
    b = rng.standard_normal(768).astype(np.float32)
    for op in ("add_inplace", "mul_inplace"):
        got = getattr(T(ca, hdev, a, [4, 768]), op)(T(ca, hdev, b, [768])).export()  # cyclic broadcast of rhs
        ref = getattr(OT(odev, a, [4, 768]), op)(OT(odev, b, [768])).export()
        assert np.array_equal(bits(got), bits(ref))
    got = T(ca, hdev, a, [4, 768]).scale_inplace(0.1767767).export()
    ref = OT(odev, a, [4, 768]).scale_inplace(np.float32(0.1767767)).export()
    assert np.array_equal(bits(got), bits(ref))
    for op in ("silu_inplace", "gelu_inplace"):
        got = getattr(T(ca, hdev, a, [4, 768]), op)().export()
        ref = getattr(OT(odev, a, [4, 768]), op)().export()
        assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("kv_f16", [False, True])
@pytest.mark.parametrize("n_heads,n_kv,hd,seq", [(6, 6, 48, 1), (6, 6, 48, 9), (32, 8, 128, 33), (8, 2, 64, 130)])
def test_attention_dot_and_combine_bitexact(ca, hdev, odev, kv_f16, n_heads, n_kv, hd, seq):
    """The runner's exact attention sequence (llama2.rs:542-590) on a pre-allocated, partially filled
    KV cache: concatenate (f32->f16), transposed-view QK^T, softmax, PV -- incl. GQA broadcast."""
    rng = np.random.default_rng(n_heads * 1000 + seq)
    cap = seq + 3
    kvt_h = ca.GGMLType.F16 if kv_f16 else ca.GGMLType.F32
    kvt_o = o.F16 if kv_f16 else o.F32
    hk = ca.HipTensor.alloc([n_kv, cap, hd], kvt_h, hdev).resize(1, 0)
    hv = ca.HipTensor.alloc([n_kv, cap, hd], kvt_h, hdev).resize(1, 0)
    ok = o.OracleTensor.alloc([n_kv, cap, hd], kvt_o, odev).resize(1, 0)
    ov = o.OracleTensor.alloc([n_kv, cap, hd], kvt_o, odev).resize(1, 0)
    for _ in range(seq):
        k = rng.standard_normal(n_kv * hd).astype(np.float32)
        v = rng.standard_normal(n_kv * hd).astype(np.float32)
        hk.concatenate(T(ca, hdev, k, [1, n_kv, hd]).transpose([1, 0, 2]), 1)
        hv.concatenate(T(ca, hdev, v, [1, n_kv, hd]).transpose([1, 0, 2]), 1)
        ok.concatenate(OT(odev, k, [1, n_kv, hd]).transpose([1, 0, 2]), 1)
        ov.concatenate(OT(odev, v, [1, n_kv, hd]).transpose([1, 0, 2]), 1)
    assert hk.shape() == [n_kv, seq, hd]
    raw_h = hk.export_raw()
    raw_o = ok.storage.view(np.uint8)
    es = 2 if kv_f16 else 4
    for h in range(n_kv):  # filled region of the cache is byte-identical (RNE f32->f16)
        lo = h * cap * hd * es
        assert np.array_equal(raw_h[lo:lo + seq * hd * es], raw_o[lo:lo + seq * hd * es])
    q = rng.standard_normal(n_heads * hd).astype(np.float32)
    scale = np.float32(1.0) / np.sqrt(np.float32(hd))
    hq = T(ca, hdev, q, [1, n_heads, hd]).transpose([1, 0, 2]).contiguous().scale_inplace(float(scale))
    oq = OT(odev, q, [1, n_heads, hd]).transpose([1, 0, 2]).contiguous().scale_inplace(scale)
    ha = hq.batch_matmul(hk.transpose([0, 2, 1]))
    oa = oq.batch_matmul(ok.transpose([0, 2, 1]))
    assert ha.shape() == [n_heads, 1, seq]
    assert np.array_equal(bits(ha.export()), bits(oa.export()))
    ha = ha.softmax_inplace(2)
    oa = oa.softmax_inplace(2)
    assert np.array_equal(bits(ha.export()), bits(oa.export()))
    hx = ha.batch_matmul(hv)
    ox = oa.batch_matmul(ov)
    assert hx.shape() == [n_heads, 1, hd]
    assert np.array_equal(bits(hx.export()), bits(ox.export()))
    assert hx.reshape([1, n_heads * hd]).shape() == [1, n_heads * hd]


def test_f16_cache_roundtrip_and_contiguous_f16(ca, hdev, odev):
    rng = np.random.default_rng(5)
    v = (rng.standard_normal(2 * 3 * 8) * 100).astype(np.float32)
    hc = ca.HipTensor.alloc([2, 3, 8], ca.GGMLType.F16, hdev).resize(1, 0)
    hc.concatenate(T(ca, hdev, v, [3, 2, 8]).transpose([1, 0, 2]), 1)
    oc = o.OracleTensor.alloc([2, 3, 8], o.F16, odev).resize(1, 0)
    oc.concatenate(OT(odev, v, [3, 2, 8]).transpose([1, 0, 2]), 1)
    assert np.array_equal(hc.export_raw(), oc.storage.view(np.uint8))
    ht = hc.transpose([1, 0, 2]).contiguous()
    ot = oc.transpose([1, 0, 2]).contiguous()
    assert np.array_equal(ht.export_raw(), ot.storage.view(np.uint8))
    with pytest.raises(ca.CrabmlError):  # F32 <- F16 is not a reference combination (concatenate.rs:69-77)
        ca.HipTensor.alloc([2, 3, 8], ca.GGMLType.F32, hdev).resize(1, 0).concatenate(hc, 1)
