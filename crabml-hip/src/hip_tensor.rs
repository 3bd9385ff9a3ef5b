use std::ptr;
use std::sync::Arc;

use crabml::bail;
use crabml::error::ErrorKind;
use crabml::error::Result;
use crabml::gguf::GGMLType;
use crabml::tensor::RopeMode;
use crabml::tensor::Tensor;
use crabml::tensor::TensorStrider;

use super::HipTensorDeviceRef;
use crate::ffi;

/// One reference on a `crabml_hip_buf_t` (a reference-counted device allocation + its GGML dtype).
/// `Arc<HipBuf>` plays the role of `Arc<wgpu::Buffer>` in crabml-wgpu/src/wgpu_tensor.rs:20-28: views share it,
/// the last clone releases the device memory back to the library's caching allocator.
pub(crate) struct HipBuf {
    pub(crate) raw: *mut ffi::crabml_hip_buf_t,
    // keeps the device alive for as long as the buffer lives (the library requires buffers to be released first)
    _device: HipTensorDeviceRef,
}

unsafe impl Send for HipBuf {}
unsafe impl Sync for HipBuf {}

impl Drop for HipBuf {
    fn drop(&mut self) {
        unsafe { ffi::crabml_hip_buf_release(self.raw) };
    }
}

#[derive(Clone)]
pub struct HipTensor {
    buf: Arc<HipBuf>,
    dtype: GGMLType,
    strider: TensorStrider,
    device: HipTensorDeviceRef,
    name: Option<String>,
}

impl HipTensor {
    /// test helper, the counterpart of `WgpuTensor::new` (crabml-wgpu/src/wgpu_tensor.rs:31-53)
    pub fn new(src: &[f32], shape: &[usize], device: HipTensorDeviceRef) -> Result<Self> {
        let strider = TensorStrider::new(shape.to_vec());
        if strider.len() != src.len() {
            bail!(ErrorKind::TensorError, "new: buffer size mismatch");
        };
        let bytes =
            unsafe { std::slice::from_raw_parts(src.as_ptr() as *const u8, std::mem::size_of_val(src)) };
        Self::from_cpu(bytes, shape, GGMLType::F32, device)
    }

    fn adopt(
        raw: *mut ffi::crabml_hip_buf_t,
        dtype: GGMLType,
        strider: TensorStrider,
        device: HipTensorDeviceRef,
    ) -> Self {
        Self {
            buf: Arc::new(HipBuf {
                raw,
                _device: device.clone(),
            }),
            dtype,
            strider,
            device,
            name: None,
        }
    }

    pub fn is_contiguous(&self) -> bool {
        self.strider.is_contiguous()
    }

    /// elements held by the underlying buffer (the capacity `resize` is bounded by)
    pub fn buf_len(&self) -> usize {
        unsafe { ffi::crabml_hip_buf_len(self.buf.raw) }
    }

    pub(crate) fn raw(&self) -> *const ffi::crabml_hip_buf_t {
        self.buf.raw
    }

    pub fn device(&self) -> &HipTensorDeviceRef {
        &self.device
    }

    fn need_contiguous(&self, op: &str) -> Result<()> {
        if !self.is_contiguous() {
            bail!(ErrorKind::TensorError, "{}: tensor is not contiguous", op);
        }
        Ok(())
    }

    /// arithmetic.rs:11-14: a[i] op= b[i % len(b)]
    fn binary_inplace(self, rhs: &Self, mul: bool) -> Result<Self> {
        let (na, nb) = (self.buf_len(), rhs.buf_len());
        if nb == 0 || na % nb != 0 {
            bail!(
                ErrorKind::TensorError,
                "lhs length is not a multiple of rhs length"
            );
        }
        if !(self.shape().last() == rhs.shape().last() || nb == 1) {
            bail!(
                ErrorKind::TensorError,
                "last dims differ: {:?} vs {:?}",
                self.shape(),
                rhs.shape()
            );
        }
        if !self.is_contiguous() || !rhs.is_contiguous() {
            bail!(ErrorKind::TensorError, "tensors must be contiguous");
        }
        let rc = unsafe {
            if mul {
                ffi::crabml_hip_mul_inplace(self.device.raw, self.buf.raw, na, rhs.buf.raw, nb)
            } else {
                ffi::crabml_hip_add_inplace(self.device.raw, self.buf.raw, na, rhs.buf.raw, nb)
            }
        };
        self.device.check(rc)?;
        Ok(self)
    }
}

impl Tensor for HipTensor {
    type DeviceRef = HipTensorDeviceRef;

    /// Uploads the GGML-layout bytes as they are stored in the GGUF file: every weight format of CpuTensorBuf -- F32, F16, Q8_0, Q4_0, Q4_1, Q5_0, Q5_1,
    /// Q2_K .. Q6_K, Q8_K.
    /// Quantized tensors are re-laid-out on the device into planes (quants | scales); no arithmetic touches them.
    fn from_cpu(
        buf: &[u8],
        shape: &[usize],
        dtype: GGMLType,
        device: Self::DeviceRef,
    ) -> Result<Self> {
        let mut raw = ptr::null_mut();
        device.check(unsafe {
            ffi::crabml_hip_buf_from_cpu(
                device.raw,
                buf.as_ptr() as *const _,
                buf.len(),
                shape.as_ptr(),
                shape.len() as i32,
                dtype as u32,
                &mut raw,
            )
        })?;
        Ok(Self::adopt(
            raw,
            dtype,
            TensorStrider::new(shape.to_vec()),
            device,
        ))
    }

    /// cpu_tensor.rs:138-165: activations and KV caches, F32 (zero filled) or F16
    fn alloc(shape: &[usize], dtype: GGMLType, device: Self::DeviceRef) -> Result<Self> {
        if dtype != GGMLType::F32 && dtype != GGMLType::F16 {
            bail!(ErrorKind::TensorError, "only f32/f16 is supported");
        }
        let n_elems = shape.iter().product::<usize>();
        let mut raw = ptr::null_mut();
        device.check(unsafe {
            ffi::crabml_hip_buf_alloc(device.raw, n_elems, dtype as u32, &mut raw)
        })?;
        Ok(Self::adopt(
            raw,
            dtype,
            TensorStrider::new(shape.to_vec()),
            device,
        ))
    }

    // ---- metadata: host side only, exactly as crabml-wgpu/src/wgpu_tensor.rs:113-187 -----------------------------

    fn resize(self, axis: usize, n: usize) -> Result<Self> {
        if axis >= self.shape().len() {
            bail!(
                ErrorKind::TensorError,
                "resize: axis {} is larger than the current shape {:?}",
                axis,
                self.shape()
            );
        }

        let mut new_shape = self.shape().to_vec();
        new_shape[axis] = n;

        let new_len: usize = new_shape.iter().product();
        if new_len > self.buf_len() {
            bail!(
                ErrorKind::TensorError,
                "resize: new shape {:?} is larger than the current shape {:?}",
                new_shape,
                self.shape()
            );
        }

        let new_strider = self.strider.resize(&new_shape)?;
        self.with_strider(new_strider)
    }

    fn dtype(&self) -> GGMLType {
        self.dtype
    }

    fn with_strider(self, strider: TensorStrider) -> Result<Self> {
        Ok(Self {
            buf: self.buf,
            dtype: self.dtype,
            strider,
            device: self.device,
            name: None,
        })
    }

    fn with_name(mut self, name: String) -> Self {
        if self.device.opts.debug_named_tensor && self.dtype == GGMLType::F32 && self.is_contiguous() {
            self.device.record_debug_tensor(name.clone(), &self);
        }

        self.name = Some(name);
        self
    }

    fn reshape(self, shape: &[usize]) -> Result<Self> {
        let strider = self.strider.reshape(shape.to_vec())?;
        self.with_strider(strider)
    }

    fn transpose(self, dims: &[usize]) -> Result<Self> {
        let strider = self.strider.transpose(dims)?;
        self.with_strider(strider)
    }

    fn strider(&self) -> &TensorStrider {
        &self.strider
    }

    fn shape(&self) -> &[usize] {
        self.strider.shape()
    }

    // ---- data movement: one FFI call each ---------------------------------------------------------------------------

    /// cpu_tensor.rs:294-304 / contiguous.rs:6-66
    fn contiguous(self) -> Result<Self> {
        if self.is_contiguous() {
            return Ok(self);
        }
        if self.dtype != GGMLType::F32 && self.dtype != GGMLType::F16 {
            bail!(ErrorKind::TensorError, "contiguous: only f32/f16");
        }
        let mut raw = ptr::null_mut();
        self.device.check(unsafe {
            ffi::crabml_hip_contiguous(
                self.device.raw,
                self.buf.raw,
                self.strider.shape().as_ptr(),
                self.strider.strides().as_ptr(),
                self.strider.dims() as i32,
                &mut raw,
            )
        })?;
        let strider = TensorStrider::new(self.shape().to_vec());
        Ok(Self::adopt(raw, self.dtype, strider, self.device.clone()))
    }

    /// cpu_tensor.rs:251-292 / concatenate.rs:12-204: the KV-cache append (f32 -> f16 rounds to nearest even)
    fn concatenate(&mut self, rhs: &Self, axis: usize) -> Result<()> {
        if self.dtype != GGMLType::F32 && self.dtype != GGMLType::F16 {
            bail!(
                ErrorKind::TensorError,
                "only f32/f16 is supported on concatenate"
            );
        }
        if rhs.dtype != GGMLType::F32 && rhs.dtype != GGMLType::F16 {
            bail!(
                ErrorKind::TensorError,
                "only f32/f16 is supported on concatenate rhs"
            );
        }
        let mismatch = rhs.strider.dims() != self.strider.dims()
            || axis >= self.strider.dims()
            || self
                .shape()
                .iter()
                .zip(rhs.shape().iter())
                .enumerate()
                .any(|(i, (a, b))| i != axis && a != b);
        if mismatch {
            bail!(
                ErrorKind::TensorError,
                "shape mismatch on concatenate, want {:?} but got {:?}",
                self.shape(),
                rhs.shape()
            );
        }
        self.device.check(unsafe {
            ffi::crabml_hip_concatenate(
                self.device.raw,
                self.buf.raw,
                self.strider.shape().as_ptr(),
                self.strider.strides().as_ptr(),
                rhs.buf.raw,
                rhs.strider.shape().as_ptr(),
                rhs.strider.strides().as_ptr(),
                self.strider.dims() as i32,
                axis as i32,
            )
        })?;

        let mut new_shape = self.strider.shape().to_vec();
        new_shape[axis] += rhs.strider.shape()[axis];
        self.strider = self.strider.resize(&new_shape)?;
        Ok(())
    }

    /// cpu_tensor.rs:306-331: rows of a (possibly quantized) table, dequantized exactly as BlockQ*::dequantize
    fn copy_rows_from(&mut self, src: &Self, src_rows: &[usize]) -> Result<()> {
        if !self.is_contiguous() {
            bail!(ErrorKind::TensorError, "dst tensor is not contiguous");
        }
        if !src.is_contiguous() {
            bail!(ErrorKind::TensorError, "src tensor is not contiguous");
        }
        if src.strider.dims() != 2 && src.strider.dims() != 1 {
            bail!(
                ErrorKind::TensorError,
                "copy_rows_from: src tensor is not 2d or 1d"
            );
        }
        let cols = *self.shape().last().unwrap();
        self.device.check(unsafe {
            ffi::crabml_hip_copy_rows_from(
                self.device.raw,
                self.buf.raw,
                src.buf.raw,
                cols,
                src_rows.as_ptr(),
                src_rows.len(),
            )
        })
    }

    /// the only call of a decode step that blocks the host (wgpu_tensor.rs:293-333 has the same contract)
    fn export(&self, dst: &mut [f32]) -> Result<()> {
        if !self.is_contiguous() {
            bail!(ErrorKind::TensorError, "export: tensor is not contiguous");
        }
        self.device.check(unsafe {
            ffi::crabml_hip_export(self.device.raw, self.buf.raw, dst.as_mut_ptr(), dst.len())
        })
    }

    /// cpu_tensor.rs:333-337: copies the WHOLE storage, keeps the shape
    fn dup(&self) -> Result<Self> {
        let mut raw = ptr::null_mut();
        self.device
            .check(unsafe { ffi::crabml_hip_dup(self.device.raw, self.buf.raw, &mut raw) })?;
        let out = Self::adopt(
            raw,
            GGMLType::F32,
            TensorStrider::new(self.shape().to_vec()),
            self.device.clone(),
        );
        if self.buf_len() != self.strider.len() {
            bail!(
                ErrorKind::TensorError,
                "invalid shape {:?} for data of length {}",
                self.shape(),
                self.buf_len()
            );
        }
        Ok(out)
    }

    // ---- compute ----------------------------------------------------------------------------------------------------

    /// rope.rs:10-45: (n_heads, head_dim) or (n_batch, n_heads, head_dim); batch row b sits at position pos + b
    fn rope_inplace(self, mode: RopeMode, pos: usize, rope_dims: usize) -> Result<Self> {
        self.need_contiguous("rope")?;
        let (n_batch, bi_stride, head_dim) = match self.strider.dims() {
            2 => (1, self.strider.len(), self.shape()[1]),
            3 => (self.shape()[0], self.strider.strides()[0], self.shape()[2]),
            _ => bail!(ErrorKind::TensorError, "rope: tensor must be 2-d or 3-d"),
        };
        let mode = match mode {
            RopeMode::Llama => 0u32,
            RopeMode::Neox => 1u32,
        };
        self.device.check(unsafe {
            ffi::crabml_hip_rope_inplace(
                self.device.raw,
                self.buf.raw,
                n_batch,
                bi_stride,
                head_dim,
                mode,
                pos,
                rope_dims,
            )
        })?;
        Ok(self)
    }

    /// rms_norm.rs:9-31
    fn rms_norm_inplace(self, eps: f32) -> Result<Self> {
        self.need_contiguous("rms_norm")?;
        let (rows, cols) = match self.shape().len() {
            1 => (1, self.shape()[0]),
            2 => (self.shape()[0], self.shape()[1]),
            _ => bail!(ErrorKind::TensorError, "rms_norm: tensor must be 1-d or 2-d"),
        };
        self.device.check(unsafe {
            ffi::crabml_hip_rms_norm_inplace(self.device.raw, self.buf.raw, rows, cols, eps)
        })?;
        Ok(self)
    }

    /// softmax.rs:11-57: last axis only; exp through the f16 table (cpu_device.rs:108-115)
    fn softmax_inplace(self, axis: usize) -> Result<Self> {
        let dims = self.strider.dims();
        if dims != 2 && dims != 3 {
            bail!(ErrorKind::TensorError, "softmax: tensor must be 2-d or 3-d");
        }
        self.need_contiguous("softmax")?;
        if axis != dims - 1 {
            bail!(
                ErrorKind::TensorError,
                "only axis={} is supported on a {} dimensions tensor",
                dims - 1,
                dims
            );
        }
        let cols = *self.shape().last().unwrap();
        let rows = if cols == 0 { 0 } else { self.strider.len() / cols };
        self.device.check(unsafe {
            ffi::crabml_hip_softmax_inplace(self.device.raw, self.buf.raw, rows, cols)
        })?;
        Ok(self)
    }

    /// silu.rs:6-13: over the whole storage, like `buf.as_f32_mut().iter_mut()`
    fn silu_inplace(self) -> Result<Self> {
        let n = self.buf_len();
        self.device
            .check(unsafe { ffi::crabml_hip_silu_inplace(self.device.raw, self.buf.raw, n) })?;
        Ok(self)
    }

    /// gelu.rs:11-22
    fn gelu_inplace(self) -> Result<Self> {
        let n = self.buf_len();
        self.device
            .check(unsafe { ffi::crabml_hip_gelu_inplace(self.device.raw, self.buf.raw, n) })?;
        Ok(self)
    }

    fn mul_inplace(self, rhs: &Self) -> Result<Self> {
        self.binary_inplace(rhs, true)
    }

    fn add_inplace(self, rhs: &Self) -> Result<Self> {
        self.binary_inplace(rhs, false)
    }

    /// cpu_tensor.rs:404-410
    fn scale_inplace(self, rhs: f32) -> Result<Self> {
        self.need_contiguous("scale")?;
        let n = self.buf_len();
        self.device.check(unsafe {
            ffi::crabml_hip_scale_inplace(self.device.raw, self.buf.raw, n, rhs)
        })?;
        Ok(self)
    }

    /// cpu_tensor.rs:371-386 / matmul_vec.rs:9-78: (m, k) @ (k,) -> (m,);  (m, k) @ (b, k) -> (b, m).
    /// `y` is quantized on the device to the weight's vec_dot_rhs_dtype (buf/api.rs:142-159), with the reference's
    /// own rounding; 16 or more rows go to the int8 matrix cores.
    fn matmul_vec(&self, y: &Self) -> Result<Self> {
        if !self.is_contiguous() || !y.is_contiguous() {
            bail!(
                ErrorKind::TensorError,
                "matmul_vec: tensors must be contiguous"
            );
        }
        if self.strider.dims() != 2 || (y.strider.dims() != 1 && y.strider.dims() != 2) {
            bail!(
                ErrorKind::TensorError,
                "matmul_vec: expect (m,k) @ (k,) or (m,k) @ (b,k)"
            );
        }
        if self.shape().last() != y.shape().last() {
            bail!(
                ErrorKind::TensorError,
                "matmul_vec: inner dims differ: {:?} vs {:?}",
                self.shape(),
                y.shape()
            );
        }
        let (m, k) = (self.shape()[0], self.shape()[1]);
        let (b, shape_c) = if y.shape().len() == 1 {
            (1, vec![m])
        } else {
            (y.shape()[0], vec![y.shape()[0], m])
        };
        let mut raw = ptr::null_mut();
        self.device.check(unsafe {
            ffi::crabml_hip_matmul_vec(self.device.raw, self.buf.raw, m, k, y.buf.raw, b, &mut raw)
        })?;
        Ok(Self::adopt(
            raw,
            GGMLType::F32,
            TensorStrider::new(shape_c),
            y.device.clone(),
        ))
    }

    /// cpu_tensor.rs:352-367 / batch_matmul.rs:15-131: (ba, m, k) @ (bb, k, n) -> (ba, m, n); the rhs is the (strided,
    /// F32 or F16) KV cache view, broadcast over the batch as the reference does (GQA)
    fn batch_matmul(&self, y: &Self) -> Result<Self> {
        if self.strider.dims() != 3 || y.strider.dims() != 3 {
            bail!(
                ErrorKind::TensorError,
                "batch_matmul: both tensors must be 3-d"
            );
        }
        if !self.is_contiguous() {
            bail!(ErrorKind::TensorError, "batch_matmul: lhs must be contiguous");
        }
        let ys = y.strider.strides();
        if !(ys[1] == 1 || ys[2] == 1) {
            bail!(
                ErrorKind::TensorError,
                "batch_matmul: rhs must be contiguous on k or n"
            );
        }
        if self.shape()[2] != y.shape()[1] {
            bail!(ErrorKind::TensorError, "batch_matmul: inner dims differ");
        }
        let (ba, m, k) = (self.shape()[0], self.shape()[1], self.shape()[2]);
        let (bb, n) = (y.shape()[0], y.shape()[2]);
        let mut raw = ptr::null_mut();
        self.device.check(unsafe {
            ffi::crabml_hip_batch_matmul(
                self.device.raw,
                self.buf.raw,
                ba,
                m,
                k,
                y.buf.raw,
                bb,
                n,
                ys[0],
                ys[1],
                ys[2],
                &mut raw,
            )
        })?;
        Ok(Self::adopt(
            raw,
            GGMLType::F32,
            TensorStrider::new(vec![ba, m, n]),
            self.device.clone(),
        ))
    }
}

#[cfg(test)]
mod tests {
    // The reference's own device tests (crabml-wgpu/src/wgpu_tensor.rs:742-1099), against this backend.  They need an
    // MI355X; the backend repository replays the same vectors through the C ABI in tests/test_hip_ops.py.
    use approx::assert_relative_eq;
    use crabml::error::Result;
    use crabml::gguf::GGMLType;
    use crabml::tensor::RopeMode;
    use crabml::tensor::Tensor;

    use super::HipTensor;
    use crate::HipTensorDevice;
    use crate::HipTensorDeviceOptions;

    #[test]
    fn test_hip_tensor_new_and_export() -> Result<()> {
        let device = HipTensorDevice::new(HipTensorDeviceOptions::new())?;
        let t1 = HipTensor::new(&[1.0, 2.0, 3.0, 4.0, 5.0, 6.0], &[2, 3], device)?;
        let mut dst = vec![0.0; 6];
        t1.export(&mut dst)?;
        assert_eq!(dst, vec![1.0, 2.0, 3.0, 4.0, 5.0, 6.0]);
        Ok(())
    }

    #[test]
    fn test_hip_tensor_add_scale_mul() -> Result<()> {
        let device = HipTensorDevice::new(HipTensorDeviceOptions::new())?;
        let t1 = HipTensor::new(&[2.0; 64], &[16, 4], device.clone())?;
        let t2 = HipTensor::new(&[3.0; 64], &[16, 4], device.clone())?;
        let t1 = t1.add_inplace(&t2)?.scale_inplace(0.5)?.mul_inplace(&t2)?;
        let mut dst = vec![0.0; 64];
        t1.export(&mut dst)?;
        assert_relative_eq!(&dst[..], &vec![7.5; 64][..], epsilon = 1e-6);
        Ok(())
    }

    #[test]
    fn test_hip_rope() -> Result<()> {
        // cpu_tensor.rs:484-500
        let device = HipTensorDevice::new(HipTensorDeviceOptions::new())?;
        let v1 = (1..=6).map(|v| v as f32).collect::<Vec<_>>();
        let t1 = HipTensor::new(&v1, &[3, 2], device)?;
        let t1 = t1.rope_inplace(RopeMode::Llama, 1, 2)?;
        let mut dst = vec![0.0; 6];
        t1.export(&mut dst)?;
        assert_relative_eq!(
            &dst[..],
            &[-1.1426396, 1.9220756, -2.7449465, 4.6856666, -4.3472533, 7.4492574][..],
            epsilon = 1e-5
        );
        Ok(())
    }

    #[test]
    fn test_hip_matmul_vec_quantized_weights() -> Result<()> {
        // a Q8_0 weight row of 32 ones with scale 1 against x = 1..32: the reference's quantize-then-dot result
        let device = HipTensorDevice::new(HipTensorDeviceOptions::new().with_strict_order(true))?;
        let mut block = vec![0u8; 34];
        block[0..2].copy_from_slice(&0x3c00u16.to_le_bytes()); // f16 1.0
        for q in block[2..].iter_mut() {
            *q = 1;
        }
        let w = HipTensor::from_cpu(&block, &[1, 32], GGMLType::Q8_0, device.clone())?;
        let x = HipTensor::new(&(1..=32).map(|v| v as f32).collect::<Vec<_>>(), &[32], device)?;
        let y = w.matmul_vec(&x)?;
        let mut dst = vec![0.0; 1];
        y.export(&mut dst)?;
        // d = 32 / 127, q = trunc(x / d): sum(q) * d
        let d = half_round(32.0f32 / 127.0);
        let sum_q: i32 = (1..=32).map(|v| (v as f32 / (32.0f32 / 127.0)) as i32).sum();
        assert_relative_eq!(dst[0], sum_q as f32 * d, epsilon = 1e-3);
        Ok(())
    }

    fn half_round(v: f32) -> f32 {
        // f32 -> f16 -> f32 (RNE), enough for the one value used above
        let bits = v.to_bits();
        let mant = bits & 0x1fff;
        let mut out = bits & !0x1fff;
        if mant > 0x1000 || (mant == 0x1000 && (bits & 0x2000) != 0) {
            out += 0x2000;
        }
        f32::from_bits(out)
    }
}
