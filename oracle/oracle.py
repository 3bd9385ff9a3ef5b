"""CPU ORACLE python face -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (crabml_oracle.c) plus a restatement of the host-side
pieces of the reference CPU backend that are pure bookkeeping:

  * TensorStrider      crabml-core/src/tensor/strider.rs
  * OracleTensor       crabml-core/src/cpu/cpu_tensor.rs   (validation + dispatch to primitives)
  * OracleDevice       crabml-core/src/cpu/cpu_device.rs
  * OracleLlamaRunner  crabml-llama2/src/llama2.rs:46-281,527-638 (Llama arch only)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (crabml_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

F32, F16, Q4_0, Q4_1, Q8_0, Q8_1, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 3, 8, 9, 12, 13, 14, 15
Q5_0, Q5_1, Q2_K, Q3_K = 6, 7, 10, 11  # gguf.rs:93-99
TYPE_NAMES = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q4_1: "Q4_1", Q8_0: "Q8_0", Q8_1: "Q8_1", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", Q8_K: "Q8_K",
              Q5_0: "Q5_0", Q5_1: "Q5_1", Q2_K: "Q2_K", Q3_K: "Q3_K"}
BLOCK_ELEMS = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q8_0: 32, Q8_1: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, Q5_0: 32, Q5_1: 32, Q2_K: 256, Q3_K: 256}
BLOCK_BYTES = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q8_0: 34, Q8_1: 36, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, Q5_0: 22, Q5_1: 24, Q2_K: 84, Q3_K: 110}
ROPE_LLAMA, ROPE_NEOX = 0, 1


class TensorError(Exception):
    """ErrorKind::TensorError (crabml-core/src/error.rs:5-33)."""


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "crabml_oracle.c")
    hdr = os.path.join(_HERE, "crabml_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, sz, u32, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)
    sig("co_f32_to_f16", C.c_uint16, f32)
    sig("co_f16_to_f32", f32, C.c_uint16)
    sig("co_f32_to_f16_vec", None, vp, vp, sz)
    sig("co_f16_to_f32_vec", None, vp, vp, sz)
    sig("co_quantize", i32, vp, sz, u32, vp)
    sig("co_dequantize", i32, vp, u32, sz, sz, vp)
    sig("co_nearest_i32", C.c_int32, f32)
    sig("co_get_scale_min_k4", None, i32, vp, vp, vp)
    sig("co_vec_dot_q8_0_q8_0", f32, vp, vp, sz)
    sig("co_vec_dot_q4_0_q8_0", f32, vp, vp, sz)
    sig("co_vec_dot_q4_1_q8_1", f32, vp, vp, sz)
    sig("co_vec_dot_q4_k_q8_k", f32, vp, vp, sz, i32, vp)
    sig("co_vec_dot_q5_k_q8_k", f32, vp, vp, sz, i32, vp)
    sig("co_vec_dot_q8_k_q8_k", f32, vp, vp, sz)
    sig("co_vec_dot_q5_0_q8_0", f32, vp, vp, sz)
    sig("co_vec_dot_q5_1_q8_1", f32, vp, vp, sz)
    sig("co_vec_dot_q2_k_q8_k", f32, vp, vp, sz, i32, vp)
    sig("co_vec_dot_q3_k_q8_k", f32, vp, vp, sz)
    sig("co_vec_dot_q6_k_q8_k", f32, vp, vp, sz)
    sig("co_vec_dot_f32_f32", f32, vp, vp, sz)
    sig("co_vec_dot_f16_f16", f32, vp, vp, sz)
    sig("co_have_avx2", i32)
    sig("co_vec_dot_q8_0_q8_0_avx2", f32, vp, vp, sz)
    sig("co_vec_dot_q4_0_q8_0_avx2", f32, vp, vp, sz)
    sig("co_vec_dot_q8_k_q8_k_avx2", f32, vp, vp, sz)
    sig("co_block_dots", i32, vp, u32, vp, sz, vp)
    sig("co_init_exp_cache", None, vp)
    sig("co_init_gelu_cache", None, vp)
    sig("co_exp_f32_cached", f32, f32, vp)
    sig("co_device_new", vp, i32, i32)
    sig("co_device_free", None, vp)
    sig("co_device_exp_cache", vp, vp)
    sig("co_matmul_vec", i32, vp, vp, u32, sz, sz, vp, sz, vp)
    sig("co_batch_matmul", i32, vp, sz, sz, sz, vp, u32, sz, sz, sz, sz, sz, vp)
    sig("co_rms_norm_inplace", None, vp, sz, sz, f32)
    sig("co_rope_inplace", None, vp, sz, sz, sz, i32, sz, sz)
    sig("co_softmax_inplace", None, vp, vp, sz, sz)
    sig("co_silu_inplace", None, vp, vp, sz)
    sig("co_gelu_inplace", None, vp, vp, sz)
    sig("co_add_inplace", None, vp, sz, vp, sz)
    sig("co_mul_inplace", None, vp, sz, vp, sz)
    sig("co_concatenate", i32, vp, u32, vp, vp, vp, u32, vp, vp, i32, i32)
    sig("co_contiguous", None, vp, vp, sz, vp, vp, i32)
    sig("co_argmax_last", sz, vp, sz)
    _lib = L
    return L


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _sz(v: Sequence[int]):
    return (C.c_size_t * len(v))(*[int(x) for x in v])


# ----------------------------------------------------------------------------------------------
# flat numpy helpers
# ----------------------------------------------------------------------------------------------
def f32_to_f16_bits(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().co_f32_to_f16_vec(_p(x), _p(out), x.size)
    return out


def f16_bits_to_f32(h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    lib().co_f16_to_f32_vec(_p(h), _p(out), h.size)
    return out


def nbytes_for(n_elems: int, typ: int) -> int:
    be = BLOCK_ELEMS[typ]
    assert n_elems % be == 0, f"{n_elems} elements is not a multiple of the {TYPE_NAMES[typ]} block ({be})"
    return n_elems // be * BLOCK_BYTES[typ]


def quantize(x: np.ndarray, typ: int) -> np.ndarray:
    """f32 -> raw block bytes (uint8 array), reference quantizers."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty(nbytes_for(x.size, typ), dtype=np.uint8)
    rc = lib().co_quantize(_p(x), x.size, typ, _p(out))
    if rc != 0:
        raise TensorError(f"quantize to {typ} is not supported")
    return out


def dequantize(raw: np.ndarray, typ: int, start: int = 0, n: Optional[int] = None) -> np.ndarray:
    raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    total = raw.size // BLOCK_BYTES[typ] * BLOCK_ELEMS[typ]
    if n is None:
        n = total - start
    out = np.empty(n, dtype=np.float32)
    rc = lib().co_dequantize(_p(raw), typ, start, n, _p(out))
    if rc != 0:
        raise TensorError(f"dequantize failed rc={rc}")
    return out


def vec_dot(w_raw: np.ndarray, wtyp: int, x_raw: np.ndarray, n_elems: int, avx2: bool = False) -> float:
    L = lib()
    w_raw = np.ascontiguousarray(w_raw).view(np.uint8)
    x_raw = np.ascontiguousarray(x_raw).view(np.uint8)
    nb = n_elems // BLOCK_ELEMS[wtyp]
    if wtyp == Q8_0:
        return (L.co_vec_dot_q8_0_q8_0_avx2 if avx2 else L.co_vec_dot_q8_0_q8_0)(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q4_0:
        return (L.co_vec_dot_q4_0_q8_0_avx2 if avx2 else L.co_vec_dot_q4_0_q8_0)(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q4_1:
        return L.co_vec_dot_q4_1_q8_1(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q4_K:
        return L.co_vec_dot_q4_k_q8_k(_p(w_raw), _p(x_raw), nb, 0, None)
    if wtyp == Q5_K:
        return L.co_vec_dot_q5_k_q8_k(_p(w_raw), _p(x_raw), nb, 0, None)
    if wtyp == Q6_K:
        return L.co_vec_dot_q6_k_q8_k(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q5_0:
        return L.co_vec_dot_q5_0_q8_0(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q5_1:
        return L.co_vec_dot_q5_1_q8_1(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q2_K:
        return L.co_vec_dot_q2_k_q8_k(_p(w_raw), _p(x_raw), nb, 0, None)
    if wtyp == Q3_K:
        return L.co_vec_dot_q3_k_q8_k(_p(w_raw), _p(x_raw), nb)
    if wtyp == Q8_K:
        return (L.co_vec_dot_q8_k_q8_k_avx2 if avx2 else L.co_vec_dot_q8_k_q8_k)(_p(w_raw), _p(x_raw), nb)
    if wtyp == F32:
        return L.co_vec_dot_f32_f32(_p(w_raw), _p(x_raw), n_elems)
    if wtyp == F16:
        return L.co_vec_dot_f16_f16(_p(w_raw), _p(x_raw), n_elems)
    raise TensorError(f"vec_dot on {wtyp}")


def q2k_overflow_count(w_raw: np.ndarray, x_raw: np.ndarray, n_elems: int) -> int:
    """products / partial sums of buf_q2_k.rs:219-222 that do not fit the reference's i16 `summs`"""
    cnt = C.c_size_t(0)
    w_raw = np.ascontiguousarray(w_raw).view(np.uint8)
    x_raw = np.ascontiguousarray(x_raw).view(np.uint8)
    lib().co_vec_dot_q2_k_q8_k(_p(w_raw), _p(x_raw), n_elems // 256, 0, C.byref(cnt))
    return int(cnt.value)


def q4k_overflow_count(w_raw: np.ndarray, x_raw: np.ndarray, n_elems: int) -> int:
    cnt = C.c_size_t(0)
    w_raw = np.ascontiguousarray(w_raw).view(np.uint8)
    x_raw = np.ascontiguousarray(x_raw).view(np.uint8)
    lib().co_vec_dot_q4_k_q8_k(_p(w_raw), _p(x_raw), n_elems // 256, 0, C.byref(cnt))
    return int(cnt.value)


def block_dots(w_raw: np.ndarray, wtyp: int, x_raw: np.ndarray, n_elems: int) -> np.ndarray:
    w_raw = np.ascontiguousarray(w_raw).view(np.uint8)
    x_raw = np.ascontiguousarray(x_raw).view(np.uint8)
    out = np.empty(n_elems // (16 if wtyp in (Q6_K, Q2_K, Q3_K) else 32), dtype=np.int32)  # Q6_K / Q2_K / Q3_K: one per 16-element scale group
    rc = lib().co_block_dots(_p(w_raw), wtyp, _p(x_raw), n_elems, _p(out))
    if rc != 0:
        raise TensorError("block_dots unsupported type")
    return out


def rhs_dtype(wtyp: int) -> int:
    return {F32: F32, F16: F16, Q8_0: Q8_0, Q4_0: Q8_0, Q8_1: Q8_1, Q4_1: Q8_1, Q8_K: Q8_K, Q4_K: Q8_K, Q5_K: Q8_K, Q6_K: Q8_K,
            Q5_0: Q8_0, Q5_1: Q8_1, Q2_K: Q8_K, Q3_K: Q8_K}[wtyp]


def argmax_last(x: np.ndarray) -> int:
    x = np.ascontiguousarray(x, dtype=np.float32)
    return int(lib().co_argmax_last(_p(x), x.size))


# ----------------------------------------------------------------------------------------------
# TensorStrider -- crabml-core/src/tensor/strider.rs
# ----------------------------------------------------------------------------------------------
class TensorStrider:
    def __init__(self, shape: Sequence[int], strides: Optional[Sequence[int]] = None):
        self._shape = [int(s) for s in shape]
        self._strides = self._compute_strides(self._shape) if strides is None else [int(s) for s in strides]

    @staticmethod
    def _compute_strides(shape):  # strider.rs:213-221
        strides = [1]
        for i in range(len(shape) - 1):
            strides.append(strides[-1] * shape[len(shape) - i - 1])
        strides.reverse()
        return strides

    def shape(self):
        return list(self._shape)

    def strides(self):
        return list(self._strides)

    def dims(self):
        return len(self._shape)

    def len(self):
        n = 1
        for s in self._shape:
            n *= s
        return n

    def resize(self, new_shape):  # strider.rs:36-51
        if len(new_shape) != len(self._shape):
            raise TensorError(f"invalid new shape {list(new_shape)} for a tensor of shape {self._shape}")
        return TensorStrider(new_shape, self._strides)

    def at(self, idx):  # strider.rs:65-86
        if len(idx) != len(self._shape):
            raise TensorError(f"invalid index {list(idx)} for tensor of shape {self._shape}")
        for i, d in enumerate(idx):
            if d >= self._shape[i]:
                raise TensorError(f"invalid index {list(idx)} for tensor of shape {self._shape}")
        return sum(d * s for d, s in zip(idx, self._strides))

    def iter(self):  # strider.rs:96-104
        pos = [0] * len(self._shape)
        out = []
        for _ in range(self.len()):
            out.append(sum(d * s for d, s in zip(pos, self._strides)))
            for i in range(len(pos) - 1, -1, -1):
                if pos[i] < self._shape[i] - 1:
                    pos[i] += 1
                    break
                pos[i] = 0
        return out

    def reshape(self, shape):  # strider.rs:143-160
        if not self.is_contiguous():
            raise TensorError("not contiguous")
        n = 1
        for s in shape:
            n *= s
        if n != self.len():
            raise TensorError(f"invalid shape {list(shape)} for a tensor's origin shape {self._shape}")
        return TensorStrider(shape)

    def transpose(self, dims):  # strider.rs:162-180
        if len(dims) != len(self._shape):
            raise TensorError(f"invalid dims {list(dims)} for a tensor of shape {self._shape}")
        return TensorStrider([self._shape[d] for d in dims], [self._strides[d] for d in dims])

    def is_contiguous(self):  # strider.rs:182-206
        if not self._strides:
            return True
        if self._strides[-1] != 1:
            return False
        last = 1
        for i in range(len(self._shape) - 1, -1, -1):
            if last != self._strides[i]:
                return False
            last *= self._shape[i]
        return True


# ----------------------------------------------------------------------------------------------
# device + tensor -- cpu_device.rs / cpu_tensor.rs
# ----------------------------------------------------------------------------------------------
class OracleDevice:
    def __init__(self, thread_num: int = 1, use_avx2: bool = False, debug_named_tensors: bool = False):
        self.thread_num = thread_num
        self.use_avx2 = use_avx2
        self.debug_named_tensors = debug_named_tensors
        self.debug_tensors = {}
        self._h = C.c_void_p(lib().co_device_new(thread_num, 1 if use_avx2 else 0))

    def dump_debug_tensor(self, name):
        v = self.debug_tensors.get(name)
        return None if v is None else v.copy()

    def __del__(self):
        try:
            if self._h:
                lib().co_device_free(self._h)
                self._h = None
        except Exception:
            pass


class OracleTensor:
    """CpuTensor restated: a flat storage array + dtype + strider.  `storage` is shared by views."""

    def __init__(self, storage: np.ndarray, dtype: int, strider: TensorStrider, device: OracleDevice,
                 owned: bool = True, name: Optional[str] = None):
        self.storage = storage  # f32 -> float32 array; f16 -> uint16 array; quantized -> uint8 raw blocks
        self._dtype = dtype
        self._strider = strider
        self.device = device
        self.owned = owned
        self.name = name

    # -- constructors -------------------------------------------------------------------------
    @staticmethod
    def new(buf, shape, device):  # cpu_tensor.rs:30-46
        buf = np.array(buf, dtype=np.float32).reshape(-1)
        n = int(np.prod(shape)) if len(shape) else 1
        if buf.size != n:
            raise TensorError(f"invalid shape {list(shape)} for data of length {buf.size}")
        return OracleTensor(buf.copy(), F32, TensorStrider(shape), device)

    @staticmethod
    def from_bytes(raw, typ, shape, device):  # cpu_tensor.rs:48-62 (zero-copy reinterpret)
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        if typ == F32:
            st = raw.view(np.float32)
        elif typ == F16:
            st = raw.view(np.uint16)
        else:
            assert raw.size % BLOCK_BYTES[typ] == 0, "data length must be a multiple of the block size"
            st = raw
        return OracleTensor(st, typ, TensorStrider(shape), device, owned=False)

    @staticmethod
    def from_cpu(raw, shape, typ, device):
        return OracleTensor.from_bytes(raw, typ, shape, device)

    @staticmethod
    def alloc(shape, dtype, device):  # cpu_tensor.rs:138-165
        if dtype not in (F32, F16):
            raise TensorError("only f32/f16 is supported")
        n = int(np.prod(shape))
        st = np.zeros(n, dtype=np.float32 if dtype == F32 else np.uint16)
        return OracleTensor(st, dtype, TensorStrider(shape), device)

    # -- metadata -----------------------------------------------------------------------------
    def dtype(self):
        return self._dtype

    def shape(self):
        return self._strider.shape()

    def strider(self):
        return self._strider

    def is_contiguous(self):
        return self._strider.is_contiguous()

    def buf_len(self):
        if self._dtype in (F32, F16):
            return self.storage.size
        return self.storage.size // BLOCK_BYTES[self._dtype] * BLOCK_ELEMS[self._dtype]

    def _view(self, strider):
        return OracleTensor(self.storage, self._dtype, strider, self.device, self.owned, None)

    def resize(self, axis, n):  # cpu_tensor.rs:167-195
        if axis >= len(self.shape()):
            raise TensorError(f"resize: axis {axis} is larger than the current shape {self.shape()}")
        ns = self.shape()
        ns[axis] = n
        if int(np.prod(ns)) > self.buf_len():
            raise TensorError(f"resize: new shape {ns} is larger than the current shape {self.shape()}")
        return self._view(self._strider.resize(ns))

    def reshape(self, shape):
        return self._view(self._strider.reshape(shape))

    def transpose(self, dims):
        return self._view(self._strider.transpose(dims))

    def with_strider(self, strider):
        return self._view(strider)

    def with_name(self, name):  # cpu_tensor.rs:232-241
        self.name = name
        if self.device.debug_named_tensors:
            self.device.debug_tensors[name] = np.array(self.storage, dtype=np.float32).copy()
        return self

    # -- data movement ------------------------------------------------------------------------
    def to_vec(self):  # cpu_tensor.rs:100-109 (test helper)
        assert self._dtype == F32
        if self.is_contiguous():
            return self.storage.copy()
        return self.storage[np.array(self._strider.iter(), dtype=np.int64)]

    def export(self):  # cpu_tensor.rs:339-349
        assert self.is_contiguous()
        return np.array(self.storage[: self._strider.len()], dtype=np.float32)

    def dup(self):  # cpu_tensor.rs:333-337: copies the WHOLE buffer, keeps the shape
        assert self._dtype == F32
        buf = self.storage.copy()
        return OracleTensor.new(buf, self.shape(), self.device)

    def contiguous(self):  # cpu_tensor.rs:294-304
        if self.is_contiguous():
            return self
        assert self._dtype in (F32, F16)
        out = OracleTensor.alloc(self.shape(), self._dtype, self.device)
        nd = self._strider.dims()
        assert nd in (2, 3)
        lib().co_contiguous(_p(self.storage), _p(out.storage), 4 if self._dtype == F32 else 2,
                            _sz(self.shape()), _sz(self._strider.strides()), nd)
        return out

    def concatenate(self, rhs, axis):  # cpu_tensor.rs:251-292
        if not self.owned:
            raise TensorError("tensor not owned on concatenate")
        if self._dtype not in (F32, F16):
            raise TensorError("only f32/f16 is supported on concatenate")
        if rhs._dtype not in (F32, F16):
            raise TensorError("only f32/f16 is supported on concatenate rhs")
        for i in range(len(self.shape())):
            if i != axis and self.shape()[i] != rhs.shape()[i]:
                raise TensorError(f"shape mismatch on concatenate, want {self.shape()} but got {rhs.shape()}")
        nd = self._strider.dims()
        rc = lib().co_concatenate(_p(self.storage), self._dtype, _sz(self.shape()), _sz(self._strider.strides()),
                                  _p(rhs.storage), rhs._dtype, _sz(rhs.shape()), _sz(rhs._strider.strides()), nd, axis)
        if rc != 0:
            raise TensorError(f"can not concatenate {self._dtype} and {rhs._dtype}")
        ns = self.shape()
        ns[axis] += rhs.shape()[axis]
        self._strider = self._strider.resize(ns)

    def copy_rows_from(self, src, rows):  # cpu_tensor.rs:306-331 + buf/api.rs:262-350
        if not self.owned:
            raise TensorError("not owned")
        if not self.is_contiguous():
            raise TensorError("dst tensor is not contiguous")
        if not src.is_contiguous():
            raise TensorError("src tensor is not contiguous")
        if src._strider.dims() not in (1, 2):
            raise TensorError("copy_rows_from: src tensor is not 2d or 1d")
        cols = self.shape()[-1]
        for dst_row, src_row in enumerate(rows):
            vals = dequantize(src.storage, src._dtype, src_row * cols, cols) if src._dtype not in (F32, F16) else (
                src.storage[src_row * cols:(src_row + 1) * cols].astype(np.float32) if src._dtype == F32
                else f16_bits_to_f32(src.storage[src_row * cols:(src_row + 1) * cols]))
            if self._dtype == F32:
                self.storage[dst_row * cols:(dst_row + 1) * cols] = vals
            else:
                self.storage[dst_row * cols:(dst_row + 1) * cols] = f32_to_f16_bits(vals)

    # -- compute ------------------------------------------------------------------------------
    def _f32(self):
        assert self._dtype == F32 and self.owned, f"not owned f32, but got {self._dtype}"
        return self.storage

    def rms_norm_inplace(self, eps):  # rms_norm.rs:9-31
        assert self.is_contiguous() and len(self.shape()) in (1, 2)
        sh = self.shape()
        rows, cols = (1, sh[0]) if len(sh) == 1 else (sh[0], sh[1])
        assert cols % 32 == 0
        lib().co_rms_norm_inplace(_p(self._f32()), rows, cols, C.c_float(eps))
        return self

    def rope_inplace(self, mode, pos, rope_dims):  # rope.rs:10-45
        assert self.is_contiguous() and self._strider.dims() in (2, 3)
        sh, st = self.shape(), self._strider.strides()
        if len(sh) == 2:
            n_batch, bi_stride, head_dim = 1, self._strider.len(), sh[1]
        else:
            n_batch, bi_stride, head_dim = sh[0], st[0], sh[2]
        lib().co_rope_inplace(_p(self._f32()), n_batch, bi_stride, head_dim, mode, pos, rope_dims)
        return self

    def softmax_inplace(self, axis):  # softmax.rs:11-57
        nd = self._strider.dims()
        assert nd in (2, 3) and self.is_contiguous()
        if axis != nd - 1:
            raise TensorError(f"only axis={nd - 1} is supported on a {nd} dimensions tensor")
        sh = self.shape()
        rows = sh[0] if nd == 2 else sh[0] * sh[1]
        lib().co_softmax_inplace(self.device._h, _p(self._f32()), rows, sh[-1])
        return self

    def silu_inplace(self):
        lib().co_silu_inplace(self.device._h, _p(self._f32()), self.storage.size)
        return self

    def gelu_inplace(self):
        lib().co_gelu_inplace(self.device._h, _p(self._f32()), self.storage.size)
        return self

    def _binary(self, rhs, fn):  # arithmetic.rs:11-14
        assert self.buf_len() % rhs.buf_len() == 0
        assert self.shape()[-1] == rhs.shape()[-1] or rhs.buf_len() == 1
        assert self.is_contiguous() and rhs.is_contiguous()
        fn(_p(self._f32()), self.storage.size, _p(rhs.storage), rhs.storage.size)
        return self

    def mul_inplace(self, rhs):
        return self._binary(rhs, lib().co_mul_inplace)

    def add_inplace(self, rhs):
        return self._binary(rhs, lib().co_add_inplace)

    def scale_inplace(self, f):  # cpu_tensor.rs:404-410
        return self.mul_inplace(OracleTensor.new([f], [1], self.device))

    def matmul_vec(self, x):  # cpu_tensor.rs:371-386 + matmul_vec.rs:9-23
        assert self.is_contiguous() and x.is_contiguous()
        assert self.shape()[-1] == x.shape()[-1]
        m, k = self.shape()[0], self.shape()[1]
        b = 1 if len(x.shape()) == 1 else x.shape()[0]
        shape_c = [m] if len(x.shape()) == 1 else [b, m]
        c = OracleTensor.alloc(shape_c, F32, self.device)
        rc = lib().co_matmul_vec(self.device._h, _p(self.storage), self._dtype, m, k, _p(x._f32()), b, _p(c.storage))
        if rc != 0:
            raise TensorError("matmul_vec: unsupported dtype/shape")
        return c

    def batch_matmul(self, b):  # cpu_tensor.rs:352-367 + batch_matmul.rs:15-45
        assert self._strider.dims() == 3 and b._strider.dims() == 3
        assert self.is_contiguous()
        bs = b._strider.strides()
        assert bs[1] == 1 or bs[2] == 1
        assert self._dtype == F32 and b._dtype in (F32, F16)
        ba, m, k = self.shape()
        bb, _, n = b.shape()
        c = OracleTensor.alloc([ba, m, n], F32, self.device)
        rc = lib().co_batch_matmul(_p(self.storage), ba, m, k, _p(b.storage), b._dtype, bb, n, bs[0], bs[1], bs[2],
                                   _p(c.storage))
        assert rc == 0
        return c


# ----------------------------------------------------------------------------------------------
# Llama runner -- crabml-llama2/src/llama2.rs (Llama architecture only)
# ----------------------------------------------------------------------------------------------
class LlamaConfig:  # model.rs:30-53
    def __init__(self, embedding_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len,
                 rms_norm_eps=1e-5, rope_dim=None):
        self.embedding_dim, self.hidden_dim, self.n_layers = embedding_dim, hidden_dim, n_layers
        self.n_heads, self.n_kv_heads, self.vocab_size, self.seq_len = n_heads, n_kv_heads, vocab_size, seq_len
        self.rms_norm_eps, self.rope_dim = rms_norm_eps, rope_dim

    def kv_dim(self):
        return self.embedding_dim * self.n_kv_heads // self.n_heads

    def head_size(self):
        return self.embedding_dim // self.n_heads


class LlamaWeights:  # model.rs:55-84 (Llama subset)
    def __init__(self):
        self.token_embed = None
        self.rms_att_weight: List = []
        self.rms_ffn_weight: List = []
        self.wq: List = []
        self.wk: List = []
        self.wv: List = []
        self.wo: List = []
        self.ffn_gate_weight: List = []
        self.ffn_down_weight: List = []
        self.ffn_up_weight: List = []
        self.rms_final_weight = None
        self.output_weight = None


class OracleLlamaRunner:
    """Llama2Runner<CpuTensor> restated (llama2.rs:46-100, 184-281, 527-638); greedy sampling."""

    def __init__(self, conf: LlamaConfig, weights: LlamaWeights, device: OracleDevice, seq_len: int,
                 use_f16_kv_cache: bool, tensor_cls=OracleTensor):
        self.T = tensor_cls
        self.conf, self.weights, self.device = conf, weights, device
        kvt = F16 if use_f16_kv_cache else F32
        self.key_cache = [self.T.alloc([conf.n_kv_heads, seq_len, conf.head_size()], kvt, device).resize(1, 0)
                          for _ in range(conf.n_layers)]
        self.value_cache = [self.T.alloc([conf.n_kv_heads, seq_len, conf.head_size()], kvt, device).resize(1, 0)
                            for _ in range(conf.n_layers)]
        self.logits = np.zeros(conf.vocab_size, dtype=np.float32)

    def kv_cache_len(self):
        return self.key_cache[0].shape()[1]

    def forward(self, tokens, pos):  # llama2.rs:184-211
        x = self.forward_llama(tokens, pos)
        x_final = self.T.alloc([self.conf.embedding_dim], F32, self.device)
        x_final.copy_rows_from(x, [len(tokens) - 1])
        ow = self.weights.output_weight if self.weights.output_weight is not None else self.weights.token_embed
        logits = ow.matmul_vec(x_final)
        self.logits = logits.export()
        return self.logits

    def forward_llama(self, tokens, pos):  # llama2.rs:213-281
        c, w, T = self.conf, self.weights, self.T
        embed_dim, n_heads, n_kv_heads, head_dim = c.embedding_dim, c.n_heads, c.n_kv_heads, c.head_size()
        rope_dim = c.rope_dim if c.rope_dim is not None else head_dim
        n_batch = len(tokens)
        x = T.alloc([n_batch, embed_dim], F32, self.device)
        x.copy_rows_from(w.token_embed, list(tokens))
        for l in range(c.n_layers):
            x_attn_orig = x.dup()
            x = x.rms_norm_inplace(c.rms_norm_eps)
            x = x.mul_inplace(w.rms_att_weight[l])
            x = x.with_name(f"attn_rmsnorm:{l}:{pos}")
            x = x.with_name(f"x_debug:{l}:{pos}")
            q = w.wq[l].matmul_vec(x)
            k = w.wk[l].matmul_vec(x)
            v = w.wv[l].matmul_vec(x)
            q = q.reshape([n_batch, n_heads, head_dim])
            k = k.reshape([n_batch, n_kv_heads, head_dim])
            q = q.rope_inplace(ROPE_LLAMA, pos, rope_dim)
            k = k.rope_inplace(ROPE_LLAMA, pos, rope_dim)
            x = self.forward_multi_query_attention(q, k, v, l, pos, n_kv_heads, n_heads, embed_dim, head_dim, n_batch)
            x = x.with_name(f"attn_out:{l}:{pos}")
            x = x.add_inplace(x_attn_orig)
            x = self.forward_ffn(x, l)
            x = x.with_name(f"ffn_out:{l}:{pos}")
        x = x.rms_norm_inplace(c.rms_norm_eps)
        x = x.mul_inplace(w.rms_final_weight)
        return x.with_name(f"final_rmsnorm:{pos}")

    def forward_multi_query_attention(self, q, k, v, l, pos, n_kv_heads, n_heads, embed_dim, head_dim, n_batch):
        # llama2.rs:527-603
        k = k.reshape([n_batch, n_kv_heads, head_dim]).transpose([1, 0, 2])
        v = v.reshape([n_batch, n_kv_heads, head_dim]).transpose([1, 0, 2])
        self.key_cache[l].concatenate(k, 1)
        self.value_cache[l].concatenate(v, 1)
        q = q.reshape([n_batch, n_heads, head_dim]).transpose([1, 0, 2]).contiguous()
        q = q.scale_inplace(np.float32(1.0) / np.sqrt(np.float32(head_dim)))
        k_cache = self.key_cache[l]
        k_orig = k_cache.strider()
        k_cache_t = k_cache.transpose([0, 2, 1])
        attn = q.batch_matmul(k_cache_t)
        attn = attn.softmax_inplace(2)
        self.key_cache[l] = k_cache_t.with_strider(k_orig)
        v_cache = self.value_cache[l]
        x_with_attn = attn.batch_matmul(v_cache)
        if n_batch == 1:
            x_with_attn = x_with_attn.reshape([n_batch, embed_dim])
        else:
            x_with_attn = x_with_attn.transpose([1, 0, 2]).contiguous().reshape([n_batch, embed_dim])
        return self.weights.wo[l].matmul_vec(x_with_attn)

    def forward_ffn(self, x, l):  # llama2.rs:605-638 (FFN norm eps is the literal 1e-5)
        w = self.weights
        x_orig = x.dup()
        x = x.rms_norm_inplace(1e-5)
        x = x.mul_inplace(w.rms_ffn_weight[l])
        h1 = w.ffn_gate_weight[l].matmul_vec(x)
        h2 = w.ffn_up_weight[l].matmul_vec(x)
        h1 = h1.silu_inplace()
        h1 = h1.mul_inplace(h2)
        x = w.ffn_down_weight[l].matmul_vec(h1)
        return x.add_inplace(x_orig)

    def generate_greedy(self, prompt_tokens, steps):
        """prefill (token at a time, llama2.rs:127-129) then `steps` greedy tokens; returns token ids."""
        base = self.kv_cache_len()
        for i, t in enumerate(prompt_tokens):
            self.forward([t], base + i)
        out = []
        tok = argmax_last(self.logits)
        out.append(tok)
        pos = self.kv_cache_len()
        for _ in range(steps - 1):
            self.forward([tok], pos)
            tok = argmax_last(self.logits)
            out.append(tok)
            pos += 1
        return out


class OracleTpLlamaRunner:
    """CPU restatement of the tensor-parallel decode step (include/crabml_hip.h, crabml_hip_llama_config_t.tp_*).

    The reference has no tensor parallelism: this is Llama2Runner::forward_llama (llama2.rs:213-281) with the
    Megatron split applied around the SAME reference ops -- every rank runs the reference's matmul_vec / rope /
    batch_matmul / softmax on its shard, the wo / ffn_down partial sums are added in rank order
    (p0 + p1 + ...) and then to the residual.  tp = 1 degenerates to OracleLlamaRunner bit for bit.
    `rank_weights[r]` holds rank r's shards (crabml_amd.tp.shard_model)."""

    def __init__(self, conf: LlamaConfig, rank_weights, device: OracleDevice, seq_len: int, use_f16_kv_cache: bool):
        self.conf, self.device, self.tp = conf, device, len(rank_weights)
        tp = self.tp
        hd = conf.head_size()
        self.local_conf = LlamaConfig(conf.n_heads // tp * hd, conf.hidden_dim // tp, conf.n_layers, conf.n_heads // tp,
                                      conf.n_kv_heads // tp, conf.vocab_size, conf.seq_len, conf.rms_norm_eps,
                                      conf.rope_dim if conf.rope_dim is not None else hd)
        self.ranks = [OracleLlamaRunner(self.local_conf, w, device, seq_len, use_f16_kv_cache) for w in rank_weights]
        self.logits = np.zeros(conf.vocab_size, dtype=np.float32)

    def kv_cache_len(self):
        return self.ranks[0].kv_cache_len()

    @staticmethod
    def _sum_in_rank_order(parts):
        s = parts[0]
        for p in parts[1:]:
            s = s.add_inplace(p)
        return s

    def forward(self, tokens, pos):
        assert len(tokens) == 1, "the tensor-parallel step decodes one token"
        c, lc, T = self.conf, self.local_conf, OracleTensor
        w0 = self.ranks[0].weights
        hd = c.head_size()
        x = T.alloc([1, c.embedding_dim], F32, self.device)
        x.copy_rows_from(w0.token_embed, list(tokens))
        for l in range(c.n_layers):
            x_orig = x.dup()
            x = x.rms_norm_inplace(c.rms_norm_eps)
            x = x.mul_inplace(w0.rms_att_weight[l])
            parts = []
            for r in self.ranks:
                w = r.weights
                q = w.wq[l].matmul_vec(x).reshape([1, lc.n_heads, hd]).rope_inplace(ROPE_LLAMA, pos, lc.rope_dim)
                k = w.wk[l].matmul_vec(x).reshape([1, lc.n_kv_heads, hd]).rope_inplace(ROPE_LLAMA, pos, lc.rope_dim)
                v = w.wv[l].matmul_vec(x)
                parts.append(r.forward_multi_query_attention(q, k, v, l, pos, lc.n_kv_heads, lc.n_heads,
                                                             lc.embedding_dim, hd, 1))
            x = self._sum_in_rank_order(parts).add_inplace(x_orig)
            x_orig = x.dup()
            x = x.rms_norm_inplace(1e-5)  # llama2.rs:611
            x = x.mul_inplace(w0.rms_ffn_weight[l])
            parts = []
            for r in self.ranks:
                w = r.weights
                h1 = w.ffn_gate_weight[l].matmul_vec(x)
                h2 = w.ffn_up_weight[l].matmul_vec(x)
                h1 = h1.silu_inplace().mul_inplace(h2)
                parts.append(w.ffn_down_weight[l].matmul_vec(h1))
            x = self._sum_in_rank_order(parts).add_inplace(x_orig)
        x = x.rms_norm_inplace(c.rms_norm_eps)
        x = x.mul_inplace(w0.rms_final_weight)
        x_final = T.alloc([c.embedding_dim], F32, self.device)
        x_final.copy_rows_from(x, [0])
        ow = w0.output_weight if w0.output_weight is not None else w0.token_embed
        if w0.output_weight is not None and ow.shape()[0] * self.tp == c.vocab_size and self.tp > 1:
            # the classifier split by vocabulary: rank r scores rows [r V / tp, (r + 1) V / tp) with the
            # reference's matmul_vec; the logits are the concatenation (every row dot is the unsplit classifier's)
            self.logits = np.concatenate([r.weights.output_weight.matmul_vec(x_final).export() for r in self.ranks])
        else:
            self.logits = ow.matmul_vec(x_final).export()
        return self.logits
