use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_char;
use std::ptr;
use std::sync::Arc;
use std::sync::Mutex;

use crabml::error::Error;
use crabml::error::ErrorKind;
use crabml::error::Result;
use crabml::tensor::Tensor;

use crate::ffi;

/// Counterpart of `WgpuTensorDeviceOptions` (crabml-wgpu/src/wgpu_device.rs:9-38).
#[derive(Debug, Clone)]
pub struct HipTensorDeviceOptions {
    /// HIP device index; one process per GPU (LOCAL_RANK) in a multi-GPU job
    pub device_ordinal: i32,

    /// record a host copy of every tensor passed through `with_name` (tests only: it blocks)
    pub debug_named_tensor: bool,

    /// CRABML_HIP_FLAG_STRICT_ORDER: matmul_vec adds the block terms in the scalar-loop order of the
    /// reference's default build; every result is then bit-identical to the CPU backend. Slow.
    pub strict_order: bool,

    /// CRABML_HIP_FLAG_PER_OP: launch every Tensor call immediately. Default (false): the calls are recorded and
    /// run at the next `export` / `sync` -- a decode token of `Llama2Runner::forward` as the fused step
    /// (five launches per layer), anything else op by op.  On a `strict_order` device the results are the same either
    /// way, bit for bit.  On the default (fast) device a token served by the fused step carries the fused step's
    /// re-associated sums (deferred 1 / rms, f32 split-KV attention past 96 positions, fused gate / up), a token that is
    /// replayed op by op -- after a mid-token `sync` / `export`, with `debug_named_tensor`, or when the call sequence
    /// deviates -- carries the per-op kernels': both inside the backend's stated tolerance of the CPU path, not equal to
    /// each other.  A host that needs one answer regardless of what it observes uses `strict_order` or `per_op`.
    pub per_op: bool,
}

impl Default for HipTensorDeviceOptions {
    fn default() -> Self {
        Self::new()
    }
}

impl HipTensorDeviceOptions {
    pub fn new() -> Self {
        Self {
            device_ordinal: 0,
            debug_named_tensor: false,
            strict_order: false,
            per_op: false,
        }
    }

    pub fn with_device_ordinal(mut self, v: i32) -> Self {
        self.device_ordinal = v;
        self
    }

    pub fn with_debug_named_tensor(mut self, v: bool) -> Self {
        self.debug_named_tensor = v;
        self
    }

    pub fn with_strict_order(mut self, v: bool) -> Self {
        self.strict_order = v;
        self
    }

    pub fn with_per_op(mut self, v: bool) -> Self {
        self.per_op = v;
        self
    }
}

pub struct HipTensorDevice {
    pub(crate) opts: HipTensorDeviceOptions,
    pub(crate) raw: *mut ffi::crabml_hip_device_t,

    /// used for test only (crabml-wgpu/src/wgpu_device.rs:47-48)
    pub debug_tensors: Mutex<HashMap<String, Vec<f32>>>,
}

// The library serialises its allocator and its error string; all work is ordered on the device's one HIP
// stream, and every entry point selects the device before it allocates or launches.
unsafe impl Send for HipTensorDevice {}
unsafe impl Sync for HipTensorDevice {}

pub type HipTensorDeviceRef = Arc<HipTensorDevice>;

/// status -> ErrorKind: the C status codes are the discriminants of crabml-core/src/error.rs:5-33, plus one
pub(crate) fn kind_from_status(rc: i32) -> ErrorKind {
    match rc {
        2 => ErrorKind::IOError,
        3 => ErrorKind::TensorNotFound,
        4 => ErrorKind::ModelError,
        5 => ErrorKind::BadInput,
        6 => ErrorKind::FormatError,
        7 => ErrorKind::TensorError,
        8 => ErrorKind::ChatTemplateNotFound,
        9 => ErrorKind::NotImplemented,
        _ => ErrorKind::Unexpected,
    }
}

impl HipTensorDevice {
    /// Counterpart of `WgpuTensorDevice::new` (crabml-wgpu/src/wgpu_device.rs:52-75). Fails when there is no
    /// usable HIP device: this backend has no CPU fallback.
    pub fn new(opts: HipTensorDeviceOptions) -> Result<HipTensorDeviceRef> {
        let abi = unsafe { ffi::crabml_hip_abi_version() };
        if abi != ffi::CRABML_HIP_ABI_VERSION {
            return Err(Error {
                kind: ErrorKind::Unexpected,
                message: format!(
                    "libcrabml_hip.so speaks ABI version {}, this crate was written against {}",
                    abi,
                    ffi::CRABML_HIP_ABI_VERSION
                ),
                cause: None,
            });
        }
        let c_opts = ffi::crabml_hip_device_options_t {
            device_ordinal: opts.device_ordinal,
            stream: ptr::null_mut(),
            flags: (if opts.strict_order {
                ffi::CRABML_HIP_FLAG_STRICT_ORDER
            } else {
                0
            }) | (if opts.per_op {
                ffi::CRABML_HIP_FLAG_PER_OP
            } else {
                0
            }),
        };
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::crabml_hip_device_create(&c_opts, &mut raw) };
        if rc != 0 || raw.is_null() {
            return Err(Error {
                kind: kind_from_status(rc),
                message: format!(
                    "crabml_hip_device_create failed for device {}: no usable HIP device (the hip backend has no cpu fallback)",
                    opts.device_ordinal
                ),
                cause: None,
            });
        }
        Ok(Arc::new(Self {
            opts,
            raw,
            debug_tensors: Mutex::new(HashMap::new()),
        }))
    }

    /// maps a C status onto `crabml::error::Error`, with the library's message for this device
    pub(crate) fn check(&self, rc: i32) -> Result<()> {
        if rc == 0 {
            return Ok(());
        }
        let mut buf = [0 as c_char; 512];
        unsafe { ffi::crabml_hip_last_error(self.raw, buf.as_mut_ptr(), buf.len()) };
        let message = unsafe { CStr::from_ptr(buf.as_ptr()) }
            .to_string_lossy()
            .into_owned();
        Err(Error {
            kind: kind_from_status(rc),
            message,
            cause: None,
        })
    }

    /// blocks until everything enqueued on the device's stream has finished
    pub fn sync(&self) -> Result<()> {
        self.check(unsafe { ffi::crabml_hip_device_sync(self.raw) })
    }

    /// bytes currently held from the HIP allocator (live + pooled)
    pub fn mem_in_use(&self) -> usize {
        unsafe { ffi::crabml_hip_device_mem_in_use(self.raw) }
    }

    pub fn record_debug_tensor(&self, name: String, tensor: &impl Tensor) {
        let mut dst = vec![0.0; tensor.strider().len()];
        tensor.export(&mut dst).unwrap();
        self.debug_tensors.lock().unwrap().insert(name, dst);
    }

    pub fn dump_debug_tensor(&self, name: &str) -> Option<Vec<f32>> {
        self.debug_tensors.lock().unwrap().get(name).cloned()
    }
}

impl Drop for HipTensorDevice {
    fn drop(&mut self) {
        // every HipTensor holds an Arc of its device, so no buffer outlives it
        unsafe { ffi::crabml_hip_device_destroy(self.raw) };
    }
}
