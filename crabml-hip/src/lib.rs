//! crabml-hip: the MI355X (gfx950) backend of crabml.
//!
//! `HipTensor` implements `crabml::tensor::Tensor` (crabml-core/src/tensor/api.rs:11-79) on top of the C ABI of
//! `libcrabml_hip.so` (include/crabml_hip.h): one FFI call per data-touching trait method, the `TensorStrider`
//! bookkeeping and the `CpuTensor` validation (crabml-core/src/cpu/cpu_tensor.rs) stay on this side of the boundary.
//! `Llama2Runner<HipTensor>` (crabml-llama2/src/llama2.rs:26-43) therefore runs unchanged; `HipLlamaRunner` is the
//! optional fast path for the Llama architecture (the same op sequence as fused kernels under one hipGraph).
//!
//! Same public surface as `crabml-wgpu` (crabml-wgpu/src/lib.rs:7-10).

mod ffi;
mod hip_device;
#[cfg(feature = "llama")]
mod hip_llama;
mod hip_tensor;

pub use hip_device::HipTensorDevice;
pub use hip_device::HipTensorDeviceOptions;
pub use hip_device::HipTensorDeviceRef;
#[cfg(feature = "llama")]
pub use hip_llama::HipLlamaRunner;
pub use hip_tensor::HipTensor;
