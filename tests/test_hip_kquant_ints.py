"""K-quants, "bit-exactly at the integer unpack level" (north_star) -- through the PRODUCTION loops.

test_hip_gemv.py::test_block_dots_bit_exact checks the raw 32- / 16-element group dots.  Here the integers that the
production kernels' float part actually consumes are dumped by those kernels themselves and compared, `==`, with the
reference's arithmetic (buf_q4_k.rs:212-263, buf_q6_k.rs:183-234) restated from the oracle's pieces:

  * single-row GEMV loops rows_partial_q4k (both header forms) / rows_partial_q6k -- the code k_gemv_q4_k, k_qkv,
    k_gemv_res_nq and k_gateup_k_lds run -- via crabml_hip_debug_superblock_ints;
  * the matrix-core GEMMs k_gemm_mfma_q4k / k_gemm_mfma_q6k via crabml_hip_debug_gemm_ints (their dump pointer set).
Per super-block: Q4_K (isum, msum) = (sum_j scale_j * sum(q4 q8), sum_j min_j * (bsum_2j + bsum_2j+1));
Q6_K isum = sum_g scale_g * sum((q6 - 32) q8)."""
import ctypes as C

import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o


def scale_min_k4(scales12):
    """get_scale_min_k4 (util.rs:19-27) for j = 0..7 -> (scales[8], mins[8])"""
    q = scales12.astype(np.int64)
    sc, mn = np.zeros(8, np.int64), np.zeros(8, np.int64)
    for j in range(8):
        if j < 4:
            sc[j], mn[j] = q[j] & 63, q[j + 4] & 63
        else:
            sc[j] = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)
            mn[j] = (q[j + 4] >> 4) | ((q[j] >> 6) << 4)
    return sc, mn


def ref_ints_q4k(w_row, xq, k):
    """per super-block (isum, msum) of one weight row against the Q8_K blocks xq"""
    nsb = k // 256
    dots = o.block_dots(w_row, o.Q4_K, xq, k).astype(np.int64).reshape(nsb, 8)
    out = np.zeros((nsb, 2), np.int64)
    for sb in range(nsb):
        blk = w_row[sb * 144:(sb + 1) * 144]
        sc, mn = scale_min_k4(blk[4:16])
        bs = xq[sb * 292 + 260:sb * 292 + 292].view(np.int16).astype(np.int64)
        out[sb, 0] = int(np.sum(sc * dots[sb]))
        out[sb, 1] = int(np.sum(mn * (bs[0::2] + bs[1::2])))
    return out


def ref_ints_q6k(w_row, xq, k):
    nsb = k // 256
    dots = o.block_dots(w_row, o.Q6_K, xq, k).astype(np.int64).reshape(nsb, 16)
    out = np.zeros((nsb, 2), np.int64)
    for sb in range(nsb):
        sc = w_row[sb * 210 + 192:sb * 210 + 208].view(np.int8).astype(np.int64)
        out[sb, 0] = int(np.sum(sc * dots[sb]))
    return out


def ref_q6k_offset_parts(w_row, xq, k):
    """sum_g scale_g * bsum_g per super-block (the term the MFMA kernel computes on its own)"""
    nsb = k // 256
    out = np.zeros(nsb, np.int64)
    for sb in range(nsb):
        sc = w_row[sb * 210 + 192:sb * 210 + 208].view(np.int8).astype(np.int64)
        bs = xq[sb * 292 + 260:sb * 292 + 292].view(np.int16).astype(np.int64)
        out[sb] = int(np.sum(sc * bs))
    return out


def make(fmt, m, k, seed, b=1):
    typ = synth.TYPE_BY_NAME[fmt]
    rng = np.random.default_rng(seed)
    raw = synth.random_blocks(rng, m * k, typ)
    x = (rng.standard_normal(b * k) * rng.uniform(0.1, 4.0)).astype(np.float32)
    return typ, raw, x


def test_reference_side_formula_reproduces_the_oracle_dot():
    """The numpy restatement above, pushed through the float part, is the oracle's vec_dot (so the integers compared on
    the GPU are the ones the reference's result is made of); get_scale_min_k4 agrees with the oracle's C function."""
    k = 1024
    for seed in range(3):
        typ, raw, x = make("Q4_K", 2, k, seed)
        xq = o.quantize(x, o.Q8_K)
        row = raw[:144 * (k // 256)]
        ints = ref_ints_q4k(row, xq, k)
        acc = 0.0
        for sb in range(k // 256):
            blk = row[sb * 144:(sb + 1) * 144]
            d, dmin = blk[0:2].view(np.float16).astype(np.float64)[0], blk[2:4].view(np.float16).astype(np.float64)[0]
            d8 = float(xq[sb * 292:sb * 292 + 4].view(np.float32)[0])
            acc += d * d8 * ints[sb, 0] - dmin * d8 * ints[sb, 1]
        ref = o.vec_dot(row, typ, xq, k)
        assert abs(acc - ref) <= 1e-5 * max(1.0, abs(ref)), (acc, ref)
        sc, mn = scale_min_k4(row[4:16])
        lib = o.lib()
        for j in range(8):
            dd, mm = C.c_uint8(0), C.c_uint8(0)
            lib.co_get_scale_min_k4(j, row[4:16].ctypes.data_as(C.c_void_p), C.byref(dd), C.byref(mm))
            assert (dd.value, mm.value) == (sc[j], mn[j])
        typ6, raw6, x6 = make("Q6_K", 2, k, seed + 10)
        xq6 = o.quantize(x6, o.Q8_K)
        row6 = raw6[:210 * (k // 256)]
        i6 = ref_ints_q6k(row6, xq6, k)
        acc = 0.0
        for sb in range(k // 256):
            d = row6[sb * 210 + 208:sb * 210 + 210].view(np.float16).astype(np.float64)[0]
            d8 = float(xq6[sb * 292:sb * 292 + 4].view(np.float32)[0])
            acc += d * d8 * i6[sb, 0]
        ref = o.vec_dot(row6, typ6, xq6, k)
        assert abs(acc - ref) <= 1e-5 * max(1.0, abs(ref)), (acc, ref)


HT = {"Q4_K": "Q4K", "Q6_K": "Q6K"}


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["Q4_K", "Q6_K"])
def test_single_row_kernels_integers_equal_the_reference(ca, hdev, fmt):
    for (m, k) in [(3, 256), (5, 768), (64, 4096), (9, 14336)]:
        typ, raw, x = make(fmt, m, k, m * 31 + k)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        hx = ca.HipTensor.new(x, [k], hdev)
        xq = o.quantize(x, o.Q8_K)
        rb = o.BLOCK_BYTES[typ] * (k // 256)
        full = w.matmul_vec(hx).export()
        for row in sorted({0, m // 2, m - 1}):
            ref = (ref_ints_q4k if fmt == "Q4_K" else ref_ints_q6k)(raw[row * rb:(row + 1) * rb], xq, k)
            for variant in ((0, 1) if fmt == "Q4_K" else (0,)):
                got, val = w.debug_superblock_ints(row, hx, variant)
                assert np.array_equal(np.asarray(got).reshape(-1, 2).astype(np.int64), ref), f"{fmt} ({m},{k}) row {row} variant {variant}"
                if variant == 0 and m < 8192:  # the dump ran the production loop: its f32 result is the GEMV kernel's
                    assert np.float32(val).view(np.uint32) == full[row].view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["Q4_K", "Q6_K"])
def test_matrix_core_gemm_integers_equal_the_reference(ca, hdev, fmt):
    for (m, k, b) in [(64, 1024, 16), (100, 2048, 37), (130, 512, 64)]:
        typ, raw, x = make(fmt, m, k, m + k + b, b=b)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        hx = ca.HipTensor.new(x, [b, k], hdev)
        ints, out = w.debug_gemm_ints(hx, b)
        nsb = k // 256
        ints = np.asarray(ints).reshape(b, m, nsb, 2).astype(np.int64)
        rb = o.BLOCK_BYTES[typ] * nsb
        plain = w.matmul_vec(hx).export().reshape(b, m)
        assert np.array_equal(np.asarray(out).reshape(b, m).view(np.uint32), plain.view(np.uint32))  # the dump run IS the product GEMM
        for bi in sorted({0, b // 2, b - 1}):
            xq = o.quantize(x[bi * k:(bi + 1) * k], o.Q8_K)
            for row in sorted({0, 17, m - 1}):
                wr = raw[row * rb:(row + 1) * rb]
                if fmt == "Q4_K":
                    assert np.array_equal(ints[bi, row], ref_ints_q4k(wr, xq, k)), (fmt, m, k, b, bi, row)
                else:
                    ref = ref_ints_q6k(wr, xq, k)[:, 0]
                    off = ref_q6k_offset_parts(wr, xq, k)
                    assert np.array_equal(ints[bi, row, :, 1], off), (fmt, m, k, b, bi, row)
                    assert np.array_equal(ints[bi, row, :, 0] - 32 * ints[bi, row, :, 1], ref), (fmt, m, k, b, bi, row)
