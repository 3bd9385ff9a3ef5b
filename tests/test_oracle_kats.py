"""Pin the CPU oracle against the reference's OWN known-answer tests.

Every test names the reference test it restates (file:line in /root/reference).  CPU only.
"""
import numpy as np
import pytest

from oracle import oracle as o


def f16(x):
    return int(o.lib().co_f32_to_f16(np.float32(x)))


def le16(v):
    return [v & 0xFF, (v >> 8) & 0xFF]


# ---------------------------------------------------------------- half crate
def test_f16_roundtrip_all_bit_patterns():
    bits = np.arange(65536, dtype=np.uint16)
    ours = o.f16_bits_to_f32(bits)
    ref = bits.view(np.float16).astype(np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(ours[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    assert np.all(np.isnan(ours[nan]))
    back = o.f32_to_f16_bits(ours[~nan])
    assert np.array_equal(back, bits[~nan])


def test_f32_to_f16_rne_matches_numpy():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-8, 6, 200000).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.9802326e-8,
                  6.1e-5, 6.097555e-5, np.inf, -np.inf], dtype=np.float32),
    ])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(o.f32_to_f16_bits(x), ref)


# ---------------------------------------------------------------- block layouts + dequant goldens
def test_q8_0_block_layout_and_dequant():  # buf_q8_0.rs:293-322
    assert o.BLOCK_BYTES[o.Q8_0] == 34
    buf = np.full(68, 1, dtype=np.uint8)
    d = le16(f16(3.0))
    buf[0:2] = d
    buf[2], buf[3], buf[4], buf[2 + 31] = 2, 3, 4, 7
    buf[34:36] = d
    buf[66], buf[67] = 9, 9
    got = o.dequantize(buf, o.Q8_0)
    exp = [6.0, 9.0, 12.0] + [3.0] * 28 + [21.0] + [3.0] * 30 + [27.0, 27.0]
    assert got.tolist() == exp


def test_q4_0_block_layout_and_dequant():  # buf_q4_0.rs:260-298
    assert o.BLOCK_BYTES[o.Q4_0] == 18
    buf = np.full(36, 1, dtype=np.uint8)
    d = le16(f16(3.0))
    for base in (0, 18):
        buf[base:base + 2] = d
        buf[base + 2], buf[base + 3], buf[base + 4] = 2, 3, 4
    got = o.dequantize(buf, o.Q4_0)
    one = [-18.0, -15.0, -12.0] + [-21.0] * 13 + [-24.0] * 16
    assert got.tolist() == one + one


def test_q4_1_block_fields_and_quantize():  # buf_q4_1.rs:287-333
    assert o.BLOCK_BYTES[o.Q4_1] == 20
    data = np.array(list(range(-8, 8)) * 2, dtype=np.float32)
    raw = o.quantize(data, o.Q4_1)
    assert o.f16_bits_to_f32(raw[0:2].view(np.uint16))[0] == 1.0
    assert o.f16_bits_to_f32(raw[2:4].view(np.uint16))[0] == -8.0
    assert raw[4:20].tolist() == [16, 50, 84, 118, 152, 186, 220, 254] * 2
    assert o.dequantize(raw, o.Q4_1).tolist() == data.tolist()  # (interleaved order round trip)


def test_q8_1_block_layout():  # buf_q8_1.rs:136-161
    assert o.BLOCK_BYTES[o.Q8_1] == 36
    buf = np.full(36, 1, dtype=np.uint8)
    buf[0:2] = le16(f16(3.0))
    buf[2:4] = le16(f16(96.0))
    buf[5], buf[6], buf[7], buf[35] = 2, 3, 4, 7
    got = o.dequantize(buf, o.Q8_1)
    assert got.tolist() == [3.0, 6.0, 9.0, 12.0] + [3.0] * 27 + [21.0]


def test_q8_k_block_layout():  # buf_q8_k.rs:233-262
    assert o.BLOCK_BYTES[o.Q8_K] == 292
    buf = np.full(292, 1, dtype=np.uint8)
    buf[0:2] = le16(f16(3.0))
    buf[2:4] = le16(f16(1.0))
    buf[4], buf[5], buf[6], buf[4 + 15], buf[283] = 2, 3, 4, 7, 10
    assert buf[0:4].view(np.float32)[0] == np.float32(0.007828236)
    assert buf[4:20].view(np.int8).tolist() == [2, 3, 4] + [1] * 12 + [7]
    bsums = buf[260:292].view(np.int16).tolist()
    assert bsums == [257] * 11 + [2561] + [257] * 4
    deq = o.dequantize(buf, o.Q8_K)
    assert deq[0] == np.float32(0.007828236) * np.float32(2.0)


def test_q8_k_quantize():  # buf_q8_k.rs:265-292
    data = np.array(list(range(-8, 8)) * 16, dtype=np.float32)
    raw = o.quantize(data, o.Q8_K)
    assert raw.size == 292
    assert raw[0:4].view(np.float32)[0] == 0.0625
    assert o.dequantize(raw, o.Q8_K).tolist() == data.tolist()
    qs = raw[4:260].view(np.int8).astype(np.int32)
    bs = raw[260:292].view(np.int16)
    assert bs.tolist() == qs.reshape(16, 16).sum(axis=1).tolist()


def test_q6_k_block_layout():  # buf_q6_k.rs:241-270
    assert o.BLOCK_BYTES[o.Q6_K] == 2 + 128 + 64 + 16 == 210
    buf = np.full(210, 1, dtype=np.uint8)
    buf[0:2] = le16(f16(3.0))
    buf[2:4] = le16(f16(1.0))
    buf[4], buf[5], buf[6], buf[4 + 15], buf[208] = 2, 3, 4, 7, 10
    # fields: ql[128] | qh[64] | scales[16] | d (f16 at 208)
    assert buf[208:210].view(np.float16)[0].astype(np.float32) == np.float32(1.5854836e-5)
    assert buf[0:16].tolist() == [0, 66, 0, 60, 2, 3, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    assert buf[128 + 48:128 + 64].tolist() == [1] * 16
    # dequantize goes through the same field offsets: element 0 = d * scales[0] * (((ql[0] & 0xF) | ((qh[0] & 3) << 4)) - 32)
    deq = o.dequantize(buf, o.Q6_K)
    d = np.float32(1.5854836e-5)
    assert deq[0] == d * np.float32(1.0) * np.float32(((0 & 0xF) | ((1 & 3) << 4)) - 32)
    assert deq[1] == d * np.float32(1.0) * np.float32(((66 & 0xF) | ((1 & 3) << 4)) - 32)


def test_q6_k_quantize_roundtrip():  # buf_q6_k.rs:283-312 (test_q6_k_quantize_2)
    data = np.array(list(range(-8, 8)) * 16, dtype=np.float32)
    raw = o.quantize(data, o.Q6_K)
    assert raw.size == 210
    assert raw[208:210].view(np.float16)[0].astype(np.float32) == np.float32(-0.001953125)
    assert o.dequantize(raw, o.Q6_K).tolist() == data.tolist()


def test_q6_k_vec_dot_q8_k_matches_dequantized_dot():
    """No value-level KAT exists for vec_dot_q6_k_q8_k (buf_q6_k.rs:183-234); the restatement is pinned through the
    pinned dequantizers: the dot equals sum(dequant(w) * dequant(x)) up to f32 re-association, and the integer
    part per 16-element scale group is exact."""
    rng = np.random.default_rng(6)
    w = (rng.standard_normal(1024) * 0.5).astype(np.float32)
    x = rng.standard_normal(1024).astype(np.float32)
    wq, xq = o.quantize(w, o.Q6_K), o.quantize(x, o.Q8_K)
    got = o.vec_dot(wq, o.Q6_K, xq, 1024)
    wd, xd = o.dequantize(wq, o.Q6_K).astype(np.float64), o.dequantize(xq, o.Q8_K).astype(np.float64)
    ref = float(wd @ xd)
    assert abs(got - ref) <= 1e-5 * float(np.abs(wd) @ np.abs(xd)) + 1e-6
    dots = o.block_dots(wq, o.Q6_K, xq, 1024)
    blocks = wq.reshape(-1, 210)
    scales = blocks[:, 192:208].view(np.int8).astype(np.float64).reshape(-1)
    dw = blocks[:, 208:210].copy().view(np.float16).astype(np.float64).reshape(-1)
    dx = xq.reshape(-1, 292)[:, 0:4].copy().view(np.float32).astype(np.float64).reshape(-1)
    per_block = (dots.astype(np.float64) * scales).reshape(-1, 16).sum(axis=1) * dw * dx
    # (the dequantized values carry one f32 rounding each: 6e-8 relative per term)
    assert abs(per_block.sum() - ref) <= 1e-7 * float(np.abs(wd) @ np.abs(xd)) + 1e-9


# ---------------------------------------------------------------- dot known answers
def _q80_blocks(qs_list, d_list):
    out = bytearray()
    for qs, d in zip(qs_list, d_list):
        out += bytes(le16(f16(d)))
        out += np.array(qs, dtype=np.int8).tobytes()
    return np.frombuffer(bytes(out), dtype=np.uint8)


@pytest.mark.parametrize("avx2", [False, True])
def test_vec_dot_q8_0_q8_0_known_answers(avx2):  # buf_q8_0.rs:325-389 (assert_eq! on f32)
    if avx2 and not o.lib().co_have_avx2():
        pytest.skip("no avx2 on this host")
    up = list(range(1, 33))
    down = list(range(32, 0, -1))
    a = _q80_blocks([up, [-v for v in up]], [0.4, 0.7])
    b = _q80_blocks([down, [-v for v in down]], [1.3, 1.4])
    assert np.float32(o.vec_dot(a, o.Q8_0, b, 64, avx2=avx2)) == np.float32(8978.046)
    a1 = _q80_blocks([up], [0.4])
    b1 = _q80_blocks([down], [1.3])
    assert np.float32(o.vec_dot(a1, o.Q8_0, b1, 32, avx2=avx2)) == np.float32(3110.453)


def _generate_data(offset, n):  # util.rs:291-297
    i = np.arange(n, dtype=np.float32)
    return (np.float32(0.1) + np.float32(2.0) * np.cos(i + np.float32(offset), dtype=np.float32)).astype(np.float32)


def test_q4_k_vec_dot_q8_k_statistical():  # buf_q4_k.rs:303-315
    a = _generate_data(0.0, 256)
    b = _generate_data(1.0, 256)
    qa = o.quantize(a, o.Q4_K)
    qb = o.quantize(b, o.Q8_K)
    dot = o.vec_dot(qa, o.Q4_K, qb, 256)
    ref = np.float32(0.0)
    for x, y in zip(a, b):
        ref = np.float32(ref + x * y)
    assert abs(ref - dot) / 256 < 0.02
    assert o.q4k_overflow_count(qa, qb, 256) == 0


def test_q5_k_block_layout_and_round_trip():  # buf_q5_k.rs:13-21 (field order qs | qh | scales | d | dmin), :334-342
    """The file's own quantize -> dequantize test: array_rmse (util.rs:300-316: sqrt(sum of squares) / n) < 0.002 on
    generate_data(0.0, 1024); pins make_qkx1_quants(32, 31, .., 9), the 6-bit scale packing, the qh bit layout and dequantize."""
    import ctypes as C

    assert o.BLOCK_BYTES[o.Q5_K] == 176 and o.lib().co_block_bytes(o.Q5_K) == 176 and o.lib().co_block_elems(o.Q5_K) == 256
    data = _generate_data(0.0, 1024)
    q = o.quantize(data, o.Q5_K)
    assert q.view(np.uint8).size == 4 * 176
    deq = o.dequantize(q, o.Q5_K, 0, 1024)
    diff = np.sqrt(np.sum((deq - data).astype(np.float32) ** 2, dtype=np.float32)) / np.float32(1024)
    assert diff < 0.002, diff
    # every level of a block uses the fifth bit somewhere: the high-bit plane is exercised
    blk = q.view(np.uint8).reshape(4, 176)
    assert np.any(blk[:, 128:160] != 0)
    # a hand-made block: d = 1, dmin = 0.5, scales[j] = j + 1, mins[j] = 1 (j < 4: plain 6-bit fields), all low nibbles 3,
    # fifth bit set for the FIRST half of chunk 0 only -> elements 0..31 = 1 * (3 + 16) - 0.5, elements 32..63 = 2 * 3 - 0.5
    b = np.zeros(176, dtype=np.uint8)
    b[0:128] = 0x33
    b[128:160] = 0x01
    b[160:164] = [1, 2, 3, 4]
    b[164:168] = [1, 1, 1, 1]
    b[172:174] = np.array([1.0], dtype=np.float16).view(np.uint8)
    b[174:176] = np.array([0.5], dtype=np.float16).view(np.uint8)
    d = o.dequantize(b, o.Q5_K, 0, 256)
    assert np.all(d[0:32] == np.float32(18.5)) and np.all(d[32:64] == np.float32(5.5))
    assert np.all(d[64:96] == np.float32(3 * 3 - 0.5)) and np.all(d[96:128] == np.float32(4 * 3 - 0.5))


def test_q5_k_vec_dot_q8_k_statistical():  # buf_q5_k.rs:344-355
    a = _generate_data(0.0, 1024)
    b = _generate_data(1.0, 1024)
    qa = o.quantize(a, o.Q5_K)
    qb = o.quantize(b, o.Q8_K)
    dot = o.vec_dot(qa, o.Q5_K, qb, 1024)
    ref = np.float32(0.0)
    for x, y in zip(a, b):
        ref = np.float32(ref + x * y)
    assert abs(ref - dot) / 1024 < 0.02
    # the dot against the dequantized weights (pinned above): the integer path agrees with the float one
    deq_a = o.dequantize(qa, o.Q5_K, 0, 1024).astype(np.float64)
    deq_b = o.dequantize(qb, o.Q8_K, 0, 1024).astype(np.float64)
    assert abs(float(deq_a @ deq_b) - dot) <= 1e-4 * float(np.abs(deq_a) @ np.abs(deq_b))
    # per-group integers of the bit-exact gate == a numpy unpack of the same bytes
    blk = qa.view(np.uint8).reshape(4, 176)
    q8 = qb.view(np.uint8).reshape(4, 292)[:, 4:260].view(np.int8).astype(np.int64)
    got = o.block_dots(qa, o.Q5_K, qb, 1024).reshape(4, 8)
    for i in range(4):
        qs, qh = blk[i, 0:128].astype(np.int64), blk[i, 128:160].astype(np.int64)
        for c in range(4):
            lo = (qs[32 * c:32 * c + 32] & 15) + 16 * ((qh >> (2 * c)) & 1)
            hi = (qs[32 * c:32 * c + 32] >> 4) + 16 * ((qh >> (2 * c + 1)) & 1)
            assert got[i, 2 * c] == int(lo @ q8[i, 64 * c:64 * c + 32]) and got[i, 2 * c + 1] == int(hi @ q8[i, 64 * c + 32:64 * c + 64])


def test_nearest_i32():  # util.rs:328-349
    cases = [(3_256_291.8, 3256292), (234_730.28, 234730), (3_271_636.3, 3271636), (143_427.25, 143427),
             (624_284.7, 624285), (601459.0, 601459), (929_129.4, 929129), (196_503.23, 196503),
             (906_489.75, 906490), (1_711_053.4, 1711053)]
    for x, e in cases:
        assert o.lib().co_nearest_i32(np.float32(x)) == e


def test_get_scale_min_k4():  # util.rs:351-359
    import ctypes as C
    data = (C.c_uint8 * 12)(*([255] * 5 + [0] * 7))
    sc, m = C.c_uint8(0), C.c_uint8(0)
    o.lib().co_get_scale_min_k4(0, data, C.byref(sc), C.byref(m))
    assert (sc.value, m.value) == (63, 63)


# ---------------------------------------------------------------- op goldens (cpu_tensor.rs:455-606)
def test_tensor_view(odev):  # cpu_tensor.rs:461-470
    t = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [2, 3], odev).reshape([3, 2])
    assert t.reshape([2, 3]).to_vec().tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]


def test_copy_from(odev):  # cpu_tensor.rs:472-488
    t1 = o.OracleTensor.new([1, 2, 3, 4], [2, 2], odev)
    t2 = o.OracleTensor.new([0, 0], [2], odev)
    t2.copy_rows_from(t1, [1])
    assert t2.to_vec().tolist() == [3.0, 4.0]
    t2.copy_rows_from(t1, [0])
    assert t2.to_vec().tolist() == [1.0, 2.0]


def test_rms_norm_formula(odev):  # cpu_tensor.rs:490-507 goldens are for the plain formula
    def simple(x):
        ss = np.float32(0.0)
        for v in x:
            ss = np.float32(ss + v * v)
        rms = np.sqrt(np.float32(ss / np.float32(len(x)) + np.float32(1e-5)))
        return [np.float32(v / rms) for v in x]

    got = simple(np.array([1, 2, 3, 4, 5, 6], dtype=np.float32))
    assert got == [np.float32(v) for v in [0.2567762, 0.5135524, 0.77032864, 1.0271049, 1.2838811, 1.5406573]]
    # the oracle primitive on a 32-multiple row agrees with the formula
    x = np.arange(1, 129, dtype=np.float32)
    t = o.OracleTensor.new(x, [128], odev).rms_norm_inplace(1e-5)
    ss = np.float32(0.0)
    for c in range(4):
        s = np.float32(0.0)
        for v in x[32 * c:32 * c + 32]:
            s = np.float32(s + v * v)
        ss = np.float32(ss + s)
    rms = np.sqrt(np.float32(ss / np.float32(128) + np.float32(1e-5)))
    assert t.to_vec().tolist() == (x / rms).tolist()


def test_rope_golden(odev):  # cpu_tensor.rs:509-527
    t = o.OracleTensor.new(np.arange(32, dtype=np.float32), [2, 16], odev)
    out = t.rope_inplace(o.ROPE_LLAMA, 1, 2).to_vec()
    exp = np.array([-0.841471, 0.54030234] + list(range(2, 16)) + [-5.6601696, 22.648676] + list(range(18, 32)),
                   dtype=np.float32)
    assert np.allclose(out, exp, rtol=0, atol=1e-5)


def test_matmul_golden(odev):  # cpu_tensor.rs:530-541
    w = o.OracleTensor.new([4.0] * 32, [16, 2], odev)
    b = o.OracleTensor.new([1.0, 2.0], [2], odev)
    assert w.matmul_vec(b).to_vec().tolist() == [12.0] * 16


def test_matmul_golden_32x8(odev):  # wgpu_tensor.rs:880-894
    w = o.OracleTensor.new(np.arange(256, dtype=np.float32), [32, 8], odev)
    b = o.OracleTensor.new([2.0] * 8, [8], odev)
    assert w.matmul_vec(b).to_vec().tolist() == [56.0 + 128.0 * i for i in range(32)]


def test_batch_matmul_golden(odev):  # wgpu_tensor.rs:897-915
    a = o.OracleTensor.new(np.arange(6, dtype=np.float32), [1, 3, 2], odev)
    b = o.OracleTensor.new([2.0, 2.0], [1, 2, 1], odev)
    assert a.batch_matmul(b).to_vec().tolist() == [2.0, 10.0, 18.0]


def test_softmax_golden(odev):  # cpu_tensor.rs:544-555 (eps 1e-3: f16 exp table)
    t = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [2, 3], odev).softmax_inplace(1)
    exp = [0.09003057, 0.24472848, 0.66524094] * 2
    assert np.allclose(t.to_vec(), exp, atol=1e-3, rtol=0)


def test_silu_golden(odev):  # cpu_tensor.rs:558-569
    t = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [6], odev).silu_inplace()
    exp = [0.7310586, 1.761594, 2.8577225, 3.928055, 4.9665356, 5.9851646]
    assert np.allclose(t.to_vec(), exp, atol=1e-1, rtol=0)
    assert np.allclose(t.to_vec(), exp, atol=2e-3, rtol=0)  # what the f16 table actually achieves


def test_gelu_golden(odev):  # wgpu_tensor.rs:1040-1056 goldens = the tanh formula
    t = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [6], odev).gelu_inplace()
    exp = [0.8411919, 1.9545977, 2.9963627, 3.99993, 5.0, 6.0]  # gelu.rs:19-22 evaluated in f32
    assert np.allclose(t.to_vec(), exp, atol=2e-3, rtol=0)


def test_contiguous_golden(odev):  # cpu_tensor.rs:572-600
    t1 = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [2, 3], odev).transpose([1, 0])
    t2 = t1.contiguous()
    assert t2.to_vec().tolist() == [1.0, 4.0, 2.0, 5.0, 3.0, 6.0] and t2.shape() == [3, 2]
    t1 = o.OracleTensor.new([1, 2, 3, 4, 5, 6], [1, 2, 3], odev).transpose([2, 1, 0])
    v1 = t1.to_vec()
    t2 = t1.contiguous()
    assert t2.to_vec().tolist() == [1.0, 4.0, 2.0, 5.0, 3.0, 6.0] and t2.shape() == [3, 2, 1]
    assert v1.tolist() == t2.to_vec().tolist()


def test_concatenate_goldens(odev):  # wgpu_tensor.rs:940-998
    t1 = o.OracleTensor.alloc([2, 2, 16], o.F32, odev).resize(0, 0)
    t1.concatenate(o.OracleTensor.new(np.arange(32, dtype=np.float32), [1, 2, 16], odev), 0)
    t1.concatenate(o.OracleTensor.new(np.arange(32, 64, dtype=np.float32), [1, 2, 16], odev), 0)
    assert t1.shape() == [2, 2, 16]
    assert t1.export().tolist() == [float(i) for i in range(64)]

    t1 = o.OracleTensor.alloc([2, 2, 16], o.F32, odev).resize(1, 0)
    t1.concatenate(o.OracleTensor.new(np.arange(32, dtype=np.float32), [2, 1, 16], odev), 1)
    t1.concatenate(o.OracleTensor.new(np.arange(32, 64, dtype=np.float32), [2, 1, 16], odev), 1)
    exp = list(range(0, 16)) + list(range(32, 48)) + list(range(16, 32)) + list(range(48, 64))
    assert t1.shape() == [2, 2, 16]
    assert t1.export().tolist() == [float(i) for i in exp]


def test_concate_2d_primitive(odev):  # concatenate.rs:211-262 (generic 2-d)
    t1 = o.OracleTensor(np.array([1, 0, 0, 4, 0, 0], dtype=np.float32), o.F32, o.TensorStrider([2, 1], [3, 1]), odev)
    t2 = o.OracleTensor(np.array([2, 5], dtype=np.float32), o.F32, o.TensorStrider([2, 1], [1, 1]), odev)
    t1.concatenate(t2, 1)
    assert t1.storage.tolist() == [1, 2, 0, 4, 5, 0] and t1.shape() == [2, 2]


# ---------------------------------------------------------------- strider suite (strider.rs:238-339)
def test_strider_suite():
    S = o.TensorStrider
    s = S([3, 4])
    assert (s.at([0, 0]), s.at([0, 3]), s.at([1, 0])) == (0, 3, 4)
    with pytest.raises(o.TensorError):
        s.reshape([4, 2])
    s = s.reshape([2, 6])
    assert (s.at([0, 0]), s.at([0, 5]), s.at([1, 0])) == (0, 5, 6)
    so = S([2, 3])
    st = so.transpose([1, 0])
    assert st.shape() == [3, 2] and st.iter() == [0, 3, 1, 4, 2, 5]
    assert st.at([1, 1]) == 4 and st.at([2, 1]) == 5
    st = st.transpose([1, 0])
    assert st.shape() == [2, 3] and st.iter() == [0, 1, 2, 3, 4, 5]
    assert S([2, 3]).is_contiguous()
    s1 = S([3, 3200])
    assert s1.strides() == [3200, 1] and s1.resize([0, 3200]).strides() == [3200, 1]
    s3 = S([3, 8, 3200])
    assert s3.strides() == [3200 * 8, 3200, 1]
    s4 = s3.resize([3, 0, 3200])
    assert s4.shape() == [3, 0, 3200] and s4.strides() == [3200 * 8, 3200, 1]


def test_argmax_returns_last_max():  # sampler.rs:109-116
    assert o.argmax_last(np.array([1, 5, 2, 5, 0], dtype=np.float32)) == 3


# ---- Q5_0 / Q5_1 / Q2_K / Q3_K: the formats the reference serves with scalar code only ------------------------------------------
_RAMP = np.array(list(range(-8, 8)) * 2, dtype=np.float32)


def _np_q5(blk, off_qh, off_qs):
    """the 32 unsigned 5-bit levels of a Q5_0 / Q5_1 block: element j < 16 = low nibble of qs[j] + bit j of qh, element j + 16 =
    high nibble + bit j + 16 (buf_q5_0.rs:26-35)"""
    qh = int.from_bytes(bytes(blk[off_qh:off_qh + 4]), "little")
    qs = blk[off_qs:off_qs + 16].astype(np.int64)
    lo = (qs & 15) | np.array([((qh >> j) & 1) << 4 for j in range(16)])
    hi = (qs >> 4) | np.array([((qh >> (j + 16)) & 1) << 4 for j in range(16)])
    return np.concatenate([lo, hi])


def test_q5_0_block_layout_quantize_and_dot():  # buf_q5_0.rs:175-218
    assert o.BLOCK_BYTES[o.Q5_0] == 22 and o.lib().co_block_bytes(o.Q5_0) == 22 and o.lib().co_block_elems(o.Q5_0) == 32
    buf = np.full(22, 1, dtype=np.uint8)  # test_q5_0_block: d = 3.0, qh = [2, 3, 4, 1], qs all 1 except qs[11] = 7
    buf[0:2] = np.array([3.0], dtype=np.float16).view(np.uint8)
    buf[2:5] = [2, 3, 4]
    buf[2 + 15] = 7
    deq = o.dequantize(buf, o.Q5_0, 0, 32)
    assert np.array_equal(deq, ((_np_q5(buf, 2, 6) - 16) * 3.0).astype(np.float32))
    b = o.quantize(_RAMP, o.Q5_0).view(np.uint8)  # test_q5_0_quantize
    assert b.size == 22 and np.float16(0.5).view(np.uint16) == b[0:2].view(np.uint16)[0]
    assert list(b[6:22]) == [0, 34, 68, 102, 136, 170, 204, 238, 0, 34, 68, 102, 136, 170, 204, 238]
    assert np.array_equal(np.round(o.dequantize(b, o.Q5_0, 0, 32)), _RAMP)
    # the dot: integers through a numpy unpack of the same bytes, then the reference's f32 expression per block
    rng = np.random.default_rng(50)
    w = o.quantize(rng.standard_normal(256).astype(np.float32), o.Q5_0).view(np.uint8).reshape(8, 22)
    x = o.quantize(rng.standard_normal(256).astype(np.float32), o.Q8_0).view(np.uint8).reshape(8, 34)
    ints = o.block_dots(w, o.Q5_0, x, 256)
    sumf = np.float32(0.0)
    for i in range(8):
        si = int((_np_q5(w[i], 2, 6) - 16) @ x[i, 2:].view(np.int8).astype(np.int64))
        assert ints[i] == si
        sumf = np.float32(sumf + np.float32(np.float32(si) * np.float32(w[i, 0:2].view(np.float16)[0])) * np.float32(x[i, 0:2].view(np.float16)[0]))
    assert np.float32(o.vec_dot(w, o.Q5_0, x, 256)) == sumf


def test_q5_1_block_layout_quantize_and_dot():  # buf_q5_1.rs:177-221
    assert o.BLOCK_BYTES[o.Q5_1] == 24 and o.lib().co_block_bytes(o.Q5_1) == 24 and o.lib().co_block_elems(o.Q5_1) == 32
    buf = np.full(24, 1, dtype=np.uint8)  # test_q5_1_block: d = 3.0, m = 1.0, qh = [2, 3, 4, 1]
    buf[0:2] = np.array([3.0], dtype=np.float16).view(np.uint8)
    buf[2:4] = np.array([1.0], dtype=np.float16).view(np.uint8)
    buf[4:7] = [2, 3, 4]
    buf[4 + 15] = 7
    deq = o.dequantize(buf, o.Q5_1, 0, 32)
    assert np.array_equal(deq, (_np_q5(buf, 4, 8) * 3.0 + 1.0).astype(np.float32))
    b = o.quantize(_RAMP, o.Q5_1).view(np.uint8)  # test_q5_1_quantize
    assert np.float32(b[0:2].view(np.float16)[0]) == np.float32(0.48388672) and b[2:4].view(np.float16)[0] == -8.0
    assert list(b[8:24]) == [0, 34, 68, 102, 136, 170, 204, 238, 17, 51, 85, 119, 153, 187, 221, 255]
    assert np.array_equal(np.round(o.dequantize(b, o.Q5_1, 0, 32)), _RAMP)
    rng = np.random.default_rng(51)
    w = o.quantize(rng.standard_normal(256).astype(np.float32), o.Q5_1).view(np.uint8).reshape(8, 24)
    x = o.quantize(rng.standard_normal(256).astype(np.float32), o.Q8_1).view(np.uint8).reshape(8, 36)
    ints = o.block_dots(w, o.Q5_1, x, 256)
    sumf = np.float32(0.0)
    for i in range(8):
        si = int(_np_q5(w[i], 4, 8) @ x[i, 4:].view(np.int8).astype(np.int64))
        assert ints[i] == si
        dd = np.float16(np.float32(w[i, 0:2].view(np.float16)[0]) * np.float32(x[i, 0:2].view(np.float16)[0]))  # f16 * f16 -> f16
        ms = np.float16(np.float32(w[i, 2:4].view(np.float16)[0]) * np.float32(x[i, 2:4].view(np.float16)[0]))
        sumf = np.float32(sumf + np.float32(np.float32(np.float32(si) * np.float32(dd)) + np.float32(ms)))
    assert np.float32(o.vec_dot(w, o.Q5_1, x, 256)) == sumf


def _np_q2k_levels(blk):
    """element order of buf_q2_k.rs:44-67: per 128-half, shift 0 / 2 / 4 / 6 of the half's 32 qs bytes"""
    qs = blk[16:80].astype(np.int64)
    return np.concatenate([(qs[32 * half:32 * half + 32] >> s) & 3 for half in range(2) for s in (0, 2, 4, 6)])


def test_q2_k_quantize_dequantize_and_dot():  # buf_q2_k.rs:260-293
    assert o.BLOCK_BYTES[o.Q2_K] == 84 and o.lib().co_block_bytes(o.Q2_K) == 84 and o.lib().co_block_elems(o.Q2_K) == 256
    a, b = _generate_data(0.0, 256), _generate_data(1.0, 256)
    qa, qb = o.quantize(a, o.Q2_K), o.quantize(b, o.Q8_K)
    dot = o.vec_dot(qa, o.Q2_K, qb, 256)
    ref = np.float32(0.0)
    for x, y in zip(a, b):
        ref = np.float32(ref + x * y)
    assert abs(ref - dot) / 256 < 0.02  # test_q2_k_vec_dot_q8_k's assertion (MAX_Q2K_PRODUCT_ERROR)
    assert o.q2k_overflow_count(qa, qb, 256) == 0
    # dequantize == an independent unpack: d * (scale & 15) * level - dmin * (scale >> 4), sixteen 16-element groups in order
    blk = qa.view(np.uint8)
    d, dmin = np.float32(blk[80:82].view(np.float16)[0]), np.float32(blk[82:84].view(np.float16)[0])
    lv = _np_q2k_levels(blk)
    want = np.empty(256, dtype=np.float32)
    for g in range(16):
        dl, ml = np.float32(d * np.float32(blk[g] & 15)), np.float32(dmin * np.float32(blk[g] >> 4))
        want[16 * g:16 * g + 16] = (dl * lv[16 * g:16 * g + 16].astype(np.float32) - ml).astype(np.float32)
    deq = o.dequantize(qa, o.Q2_K, 0, 256)
    assert np.array_equal(deq, want)
    q8 = qb.view(np.uint8)[4:260].view(np.int8).astype(np.int64)
    ints = o.block_dots(qa, o.Q2_K, qb, 256)
    assert [int(v) for v in ints] == [int(lv[16 * g:16 * g + 16] @ q8[16 * g:16 * g + 16]) for g in range(16)]
    # the dot from those integers, with the reference's single f32 expression per super-block (buf_q2_k.rs:255)
    bs = qb.view(np.uint8)[260:292].view(np.int16).astype(np.int64)
    isum = sum(int(blk[g] & 15) * int(ints[g]) for g in range(16))
    summs = sum(int(bs[g]) * int(blk[g] >> 4) for g in range(16))
    d8 = qb.view(np.uint8)[0:4].view(np.float32)[0]
    assert np.float32(dot) == np.float32(np.float32(np.float32(d8 * d) * np.float32(isum)) - np.float32(np.float32(d8 * dmin) * np.float32(summs)))


def test_q2_k_quantizer_reads_the_first_block_for_every_block():  # buf_q2_k.rs:197 (`data`, not `data_chunk`)
    a, b = _generate_data(0.0, 256), _generate_data(0.5, 256) * np.float32(0.7)
    two = o.quantize(np.concatenate([a, b]), o.Q2_K).view(np.uint8).reshape(2, 84)
    blk = two[1]
    d, dmin = np.float32(blk[80:82].view(np.float16)[0]), np.float32(blk[82:84].view(np.float16)[0])
    lv = _np_q2k_levels(blk)
    n_checked = 0
    for g in range(16):
        dl = np.float32(d * np.float32(blk[g] & 15))
        if dl == 0:
            continue  # the levels of make_qkx1_quants stay (buf_q2_k.rs:192-194)
        dm = np.float32(dmin * np.float32(blk[g] >> 4))
        from_first = [min(3, max(0, o.lib().co_nearest_i32(np.float32(np.float32(a[16 * g + i] + dm) / dl)))) for i in range(16)]
        from_own = [min(3, max(0, o.lib().co_nearest_i32(np.float32(np.float32(b[16 * g + i] + dm) / dl)))) for i in range(16)]
        assert list(lv[16 * g:16 * g + 16]) == from_first  # the second block's levels come from the FIRST block's values
        n_checked += from_first != from_own
    assert n_checked >= 8  # ... and that is visible: they are not what the block's own values give


def _np_q3k(blk):
    """(levels -4..3 in element order, the 16 scales - 32) of a Q3_K block: buf_q3_k.rs:37-88"""
    hm, qs, sc = blk[0:32].astype(np.int64), blk[32:96].astype(np.int64), blk[96:108].astype(np.int64)
    lv = []
    for half in range(2):
        for si, s in enumerate((0, 2, 4, 6)):
            bit = (hm >> (4 * half + si)) & 1
            lv.append(((qs[32 * half:32 * half + 32] >> s) & 3) - 4 * (1 - bit))
    scales = [int((sc[j] & 15) if j < 8 else (sc[j - 8] >> 4)) | (int((sc[8 + j % 4] >> (2 * (j // 4))) & 3) << 4) for j in range(16)]
    return np.concatenate(lv), np.array(scales) - 32


def test_q3_k_quantize_dequantize_and_dot():  # buf_q3_k.rs:331-365 (its assertions are commented out there: pinned here instead)
    assert o.BLOCK_BYTES[o.Q3_K] == 110 and o.lib().co_block_bytes(o.Q3_K) == 110 and o.lib().co_block_elems(o.Q3_K) == 256
    a, b = _generate_data(0.0, 512), _generate_data(1.0, 512)
    qa, qb = o.quantize(a, o.Q3_K), o.quantize(b, o.Q8_K)
    deq = o.dequantize(qa, o.Q3_K, 0, 512)
    rmse = np.sqrt(np.sum((deq - a) ** 2)) / 512
    assert rmse < 0.02, rmse  # 3-bit levels of a cosine of amplitude 2: a sanity bound on make_q3_quants + the packing
    blks = qa.view(np.uint8).reshape(2, 110)
    q8 = qb.view(np.uint8).reshape(2, 292)
    ints = o.block_dots(qa, o.Q3_K, qb, 512).reshape(2, 16)
    sums = np.zeros(8, dtype=np.float32)
    for i in range(2):
        lv, sc = _np_q3k(blks[i])
        d = np.float32(blks[i, 108:110].view(np.float16)[0])
        want = np.concatenate([np.float32(d * np.float32(sc[g])) * lv[16 * g:16 * g + 16].astype(np.float32) for g in range(16)]).astype(np.float32)
        assert np.array_equal(deq[256 * i:256 * i + 256], want)
        x8 = q8[i, 4:260].view(np.int8).astype(np.int64)
        assert [int(v) for v in ints[i]] == [int(lv[16 * g:16 * g + 16] @ x8[16 * g:16 * g + 16]) for g in range(16)]
        # eight i32 lanes per block (element e feeds lane e % 8), eight f32 sums across blocks: buf_q3_k.rs:303-327
        aux32 = np.zeros(8, dtype=np.int64)
        for g in range(16):
            p = lv[16 * g:16 * g + 16] * x8[16 * g:16 * g + 16] * sc[g]
            aux32 += p[0:8] + p[8:16]
        dd = np.float32(d * q8[i, 0:4].view(np.float32)[0])
        sums = (sums + dd * aux32.astype(np.float32)).astype(np.float32)
    r = sums[0]
    for l in range(1, 8):
        r = np.float32(r + sums[l])
    assert np.float32(o.vec_dot(qa, o.Q3_K, qb, 512)) == r
    # a hand-made block: d = 2, every scale field 33 (-> +1), low bits 1, hmask bit 0 set for the first 16 positions only:
    # elements 0..15 = 2 * 1 * (1 - 0), elements 16..31 = 2 * (1 - 4); elements 32.. (hmask bit 1 clear) = 2 * (1 - 4)
    hb = np.zeros(110, dtype=np.uint8)
    hb[0:16] = 0x01
    hb[32:96] = 0x55
    hb[96:104] = 0x11
    hb[104:108] = 0xAA
    hb[108:110] = np.array([2.0], dtype=np.float16).view(np.uint8)
    hd = o.dequantize(hb, o.Q3_K, 0, 256)
    assert np.all(hd[0:16] == 2.0) and np.all(hd[16:256] == -6.0)
