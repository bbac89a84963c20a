//! Drop-in façade with the reference's API (`DenoiseState`, `RnnModel`, jneem/nnnoiseless src/denoise.rs,
//! src/rnn.rs) over the MI355X backend's C ABI (include/nnn_batch.h).  Written against the FFI only;
//! it was not compiled in the build image, which has no Rust toolchain.
use std::os::raw::{c_float, c_int, c_void};

#[repr(C)]
pub struct RawModel {
    _p: [u8; 0],
}
#[repr(C)]
pub struct RawBatch {
    _p: [u8; 0],
}
#[repr(C)]
pub struct RawNode {
    _p: [u8; 0],
}

extern "C" {
    fn nnn_model_from_bytes(bytes: *const u8, len: usize) -> *mut RawModel;
    fn nnn_model_default() -> *mut RawModel;
    fn nnn_model_free(m: *mut RawModel);
    fn nnn_model_clone(m: *const RawModel) -> *mut RawModel;
    fn nnn_node_create(model: *const RawModel, n_streams: c_int, devices: *const c_int, n_devices: c_int, opts: *const BatchOpts) -> *mut RawNode;
    fn nnn_node_destroy(n: *mut RawNode);
    fn nnn_node_reset(n: *mut RawNode) -> c_int;
    fn nnn_node_process_host(
        n: *mut RawNode,
        input: *const c_float,
        output: *mut c_float,
        vad: *mut c_float,
        n_frames: c_int,
        stream_stride: usize,
        frame_stride: usize,
    ) -> c_int;
    fn nnn_node_num_shards(n: *const RawNode) -> c_int;
    fn nnn_node_shard_cpus(n: *const RawNode, i: c_int) -> *const std::os::raw::c_char;
    fn nnn_node_synchronize(n: *mut RawNode) -> c_int;
    fn nnn_node_process_device_streams(
        n: *mut RawNode,
        d_in: *const *const c_float,
        d_out: *const *mut c_float,
        d_vad: *const *mut c_float,
        hip_streams: *const *mut c_void,
        n_tables: c_int,
        n_frames: c_int,
        stream_stride: usize,
        frame_stride: usize,
    ) -> c_int;
    fn nnn_batch_create(model: *const RawModel, n_streams: c_int, device: c_int) -> *mut RawBatch;
    fn nnn_batch_destroy(b: *mut RawBatch);
    fn nnn_batch_reset(b: *mut RawBatch) -> c_int;
    fn nnn_batch_clone(b: *mut RawBatch) -> *mut RawBatch;
    fn nnn_batch_process_host(
        b: *mut RawBatch,
        input: *const c_float,
        output: *mut c_float,
        vad: *mut c_float,
        n_frames: c_int,
        stream_stride: usize,
        frame_stride: usize,
    ) -> c_int;
    fn nnn_batch_process_device(
        b: *mut RawBatch,
        d_in: *const c_float,
        d_out: *mut c_float,
        d_vad: *mut c_float,
        n_frames: c_int,
        stream_stride: usize,
        frame_stride: usize,
        hip_stream: *mut c_void,
    ) -> c_int;
    fn nnn_batch_fault(b: *const RawBatch) -> c_int;
    fn nnn_host_alloc(bytes: usize) -> *mut c_void;
    fn nnn_host_free(p: *mut c_void);
    fn nnn_model_from_rnnoise_text(text: *const u8, len: usize) -> *mut RawModel;
    fn nnn_batch_create_grouped(
        models: *const *const RawModel,
        group_streams: *const c_int,
        n_groups: c_int,
        device: c_int,
    ) -> *mut RawBatch;
    fn nnn_batch_create_opts(
        models: *const *const RawModel,
        group_streams: *const c_int,
        n_groups: c_int,
        device: c_int,
        opts: *const BatchOpts,
    ) -> *mut RawBatch;
    fn nnn_batch_device_bytes(b: *const RawBatch) -> usize;
    fn nnn_batch_process_pcm_host(
        b: *mut RawBatch,
        input: *const c_void,
        output: *mut c_void,
        vad: *mut c_float,
        n_frames: c_int,
        layout: *const PcmLayout,
    ) -> c_int;
}

/// `struct nnn_pcm_layout` (include/nnn_batch.h): the sample formats and channel interleave of the reference's callers
/// (src/nnnoiseless.rs:147-177,301-331, src/signal.rs:95-100,123-127), converted inside the first and last kernels.
#[repr(C)]
pub struct PcmLayout {
    pub format: c_int,        // 0 = f32 in i16 range, 1 = i16, 2 = f32 in [-1, 1]
    pub channels: c_int,
    pub discard_first: c_int, // drop the first frame after new()/reset, as the CLI and DenoiseSignal do
    pub reserved: c_int,
    pub group_stride: usize,  // elements between groups of `channels` interleaved streams
    pub frame_stride: usize,  // elements between frames of a group (>= 480 * channels)
}

/// Model parameters; `from_bytes` returns `None` exactly where the reference does (src/rnn.rs:196-222).
pub struct RnnModel(*mut RawModel);
unsafe impl Send for RnnModel {}
unsafe impl Sync for RnnModel {}

impl RnnModel {
    pub fn from_bytes(bytes: &[u8]) -> Option<RnnModel> {
        let p = unsafe { nnn_model_from_bytes(bytes.as_ptr(), bytes.len()) };
        if p.is_null() {
            None
        } else {
            Some(RnnModel(p))
        }
    }
    pub fn from_static_bytes(bytes: &'static [u8]) -> Option<RnnModel> {
        Self::from_bytes(bytes)
    }
    /// An RNNoise text model (what train/convert_rnnoise.py converts first).
    pub fn from_rnnoise_text(text: &str) -> Option<RnnModel> {
        let p = unsafe { nnn_model_from_rnnoise_text(text.as_ptr(), text.len()) };
        if p.is_null() {
            None
        } else {
            Some(RnnModel(p))
        }
    }
}
impl Default for RnnModel {
    fn default() -> RnnModel {
        RnnModel(unsafe { nnn_model_default() })
    }
}
/// `#[derive(Clone)]` in the reference (src/rnn.rs:54): an independent copy of the parameters (`nnn_model_clone`).
impl Clone for RnnModel {
    fn clone(&self) -> RnnModel {
        let p = unsafe { nnn_model_clone(self.0 as *const RawModel) };
        assert!(!p.is_null(), "nnnoiseless-mi355x: model clone failed");
        RnnModel(p)
    }
}
impl Drop for RnnModel {
    fn drop(&mut self) {
        unsafe { nnn_model_free(self.0) }
    }
}

/// `n` independent denoisers advanced in lock-step (the loop of src/signal.rs:102-104 as one call).
pub struct BatchDenoiser {
    raw: *mut RawBatch,
    n: usize,
}
unsafe impl Send for BatchDenoiser {}
// The reference asserts `DenoiseState: Send + Sync` (src/denoise.rs:125).  Every call that touches the batch's state takes
// `&mut self`; what `&self` reaches (`fault`, `device_bytes`) only reads fields the backend never changes after creation or
// a word of page-locked memory the device writes: sharing a `&BatchDenoiser` between threads is sound.
unsafe impl Sync for BatchDenoiser {}

/// `struct nnn_batch_opts` (include/nnn_batch.h)
#[repr(C)]
#[derive(Default)]
pub struct BatchOpts {
    pub max_group_frames: c_int,
    pub reserved: [c_int; 7],
}

impl BatchDenoiser {
    /// A batch sized for calls of at most `max_group_frames` frames: a real-time host that ticks one frame per call (what
    /// `DenoiseSignal` does, src/signal.rs:102-104) passes 1 and holds 33 KB per stream instead of 360.
    pub fn sized(n_streams: usize, max_group_frames: usize, model: Option<&RnnModel>, device: i32) -> Option<BatchDenoiser> {
        let m = model.map_or(std::ptr::null(), |m| m.0 as *const RawModel);
        let n = n_streams as c_int;
        let opts = BatchOpts { max_group_frames: max_group_frames as c_int, ..Default::default() };
        let raw = unsafe { nnn_batch_create_opts(&m, &n, 1, device, &opts) };
        if raw.is_null() {
            None
        } else {
            Some(BatchDenoiser { raw, n: n_streams })
        }
    }
    pub fn device_bytes(&self) -> usize {
        unsafe { nnn_batch_device_bytes(self.raw) }
    }
    pub fn new(n_streams: usize, model: Option<&RnnModel>, device: i32) -> Option<BatchDenoiser> {
        let m = model.map_or(std::ptr::null(), |m| m.0 as *const RawModel);
        let raw = unsafe { nnn_batch_create(m, n_streams as c_int, device) };
        if raw.is_null() {
            None
        } else {
            Some(BatchDenoiser { raw, n: n_streams })
        }
    }
    /// `input`/`output`: `[n_streams][n_frames][480]`; `vad`: `[n_frames][n_streams]`.
    pub fn process(&mut self, output: &mut [f32], input: &[f32], vad: &mut [f32], n_frames: usize) {
        assert_eq!(input.len(), self.n * n_frames * DenoiseState::FRAME_SIZE);
        assert_eq!(output.len(), input.len());
        assert_eq!(vad.len(), self.n * n_frames);
        let rc = unsafe {
            nnn_batch_process_host(
                self.raw,
                input.as_ptr(),
                output.as_mut_ptr(),
                vad.as_mut_ptr(),
                n_frames as c_int,
                n_frames * DenoiseState::FRAME_SIZE,
                DenoiseState::FRAME_SIZE,
            )
        };
        assert_eq!(rc, 0, "nnnoiseless-mi355x: backend error");
    }
    /// Device-resident buffers (see include/nnn_batch.h); asynchronous.
    pub unsafe fn process_device(
        &mut self,
        d_out: *mut f32,
        d_in: *const f32,
        d_vad: *mut f32,
        n_frames: usize,
        stream_stride: usize,
        frame_stride: usize,
        hip_stream: *mut c_void,
    ) {
        let rc = nnn_batch_process_device(self.raw, d_in, d_out, d_vad, n_frames as c_int, stream_stride, frame_stride, hip_stream);
        assert_eq!(rc, 0, "nnnoiseless-mi355x: backend error");
    }
    /// True once a frame hand-off inside the pitch stage has failed (sticky until `reset`): for hosts that drive
    /// `nnn_batch_process_device` on their own HIP stream and synchronise that stream themselves.
    pub fn fault(&self) -> bool {
        unsafe { nnn_batch_fault(self.raw) != 0 }
    }

    pub fn reset(&mut self) {
        unsafe { nnn_batch_reset(self.raw) };
    }
}
/// `DenoiseState` is `Clone` in the reference (src/denoise.rs:36): a second batch with the same models and a device-side
/// copy of every stream's state (`nnn_batch_clone`).
impl Clone for BatchDenoiser {
    fn clone(&self) -> BatchDenoiser {
        let raw = unsafe { nnn_batch_clone(self.raw) };
        assert!(!raw.is_null(), "nnnoiseless-mi355x: clone failed");
        BatchDenoiser { raw, n: self.n }
    }
}
impl Drop for BatchDenoiser {
    fn drop(&mut self) {
        unsafe { nnn_batch_destroy(self.raw) }
    }
}

/// Same surface as `nnnoiseless::DenoiseState<'model>` (src/denoise.rs:36-116), lifetime and `Clone` included: host code that names
/// `DenoiseState<'static>` or borrows a model for `'model` compiles unchanged.  The backend copies the model to the device when the
/// state is made, so the borrow is only a marker (`PhantomData`): it keeps the reference's rule that a borrowed model outlives the
/// states made from it, without holding a pointer to it.
#[derive(Clone)]
pub struct DenoiseState<'model> {
    batch: BatchDenoiser,
    _model: std::marker::PhantomData<&'model RnnModel>,
}

impl DenoiseState<'static> {
    /// A `DenoiseState` processes this many samples at a time.
    pub const FRAME_SIZE: usize = 480;

    /// `DenoiseState::new()` (src/denoise.rs:53-55): the built-in model.
    pub fn new() -> Box<DenoiseState<'static>> {
        Box::new(DenoiseState { batch: BatchDenoiser::sized(1, 1, None, 0).expect("no MI355X backend"), _model: std::marker::PhantomData })
    }
    /// `DenoiseState::from_model(model)` (src/denoise.rs:61-63): the state owns the model (here: its device copy; the host copy is dropped).
    pub fn from_model(model: RnnModel) -> Box<DenoiseState<'static>> {
        Box::new(DenoiseState { batch: BatchDenoiser::sized(1, 1, Some(&model), 0).expect("no MI355X backend"), _model: std::marker::PhantomData })
    }
}

impl<'model> DenoiseState<'model> {
    /// `DenoiseState::with_model(&model)` (src/denoise.rs:72-74): the same model shared between states.
    pub fn with_model(model: &'model RnnModel) -> Box<DenoiseState<'model>> {
        Box::new(DenoiseState { batch: BatchDenoiser::sized(1, 1, Some(model), 0).expect("no MI355X backend"), _model: std::marker::PhantomData })
    }
    /// src/denoise.rs:95-116: 480 samples in the range of an i16 in, 480 out, returns the voice-activity probability.
    pub fn process_frame(&mut self, output: &mut [f32], input: &[f32]) -> f32 {
        assert!(input.len() == DenoiseState::FRAME_SIZE); // src/features.rs:98
        let mut vad = [0.0f32];
        self.batch.process(&mut output[..DenoiseState::FRAME_SIZE], input, &mut vad, 1);
        vad[0]
    }
}

/// `nnnoiseless::DenoiseSignal` (src/signal.rs:29-138) over one batch: the per-channel loop of `refill_out_bufs`
/// (src/signal.rs:102-104) is one call on a batch of `CHANNELS` streams.  Needs the `dasp` feature, like the reference's.
#[cfg(feature = "dasp")]
pub mod signal {
    use super::{BatchDenoiser, DenoiseState, RnnModel};
    use dasp::frame::Frame;
    use dasp::sample::Sample;
    use dasp::signal::Signal;

    const FRAME_SIZE: usize = 480;

    #[derive(Clone)]
    pub struct DenoiseSignal<'model, S: Signal> {
        input: S,
        states: BatchDenoiser,               // CHANNELS streams in lock-step
        in_bufs: Vec<f32>,                   // [channel][480]
        out_bufs: Vec<f32>,
        vad: Vec<f32>,
        out_idx: usize,
        _model: std::marker::PhantomData<&'model RnnModel>,
    }

    impl<'model, S: Signal> DenoiseSignal<'model, S> {
        fn make(input: S, model: Option<&RnnModel>) -> Self {
            let ch = S::Frame::CHANNELS;
            DenoiseSignal {
                input,
                states: BatchDenoiser::sized(ch, 1, model, 0).expect("no MI355X backend"),
                in_bufs: vec![0.0; ch * FRAME_SIZE],
                out_bufs: vec![0.0; ch * FRAME_SIZE],
                vad: vec![0.0; ch],
                out_idx: 0,
                _model: std::marker::PhantomData,
            }
            .discard_first_frame()
        }
        /// src/signal.rs:39-48
        pub fn new(input: S) -> DenoiseSignal<'static, S> {
            DenoiseSignal::<'static, S>::make(input, None)
        }
        /// src/signal.rs:55-64
        pub fn with_model(input: S, model: &'model RnnModel) -> DenoiseSignal<'model, S> {
            Self::make(input, Some(model))
        }
        /// src/signal.rs:72-81
        pub fn from_model(input: S, model: RnnModel) -> DenoiseSignal<'static, S> {
            DenoiseSignal::<'static, S>::make(input, Some(&model))
        }
        fn discard_first_frame(mut self) -> Self {
            self.refill_out_bufs();
            self.refill_out_bufs();
            self
        }
        /// Returns true if the input was not exhausted (src/signal.rs:90-106).
        fn refill_out_bufs(&mut self) -> bool {
            if self.input.is_exhausted() {
                return false;
            }
            for i in 0..FRAME_SIZE {
                for (ch, samp) in self.input.next().to_float_frame().channels().enumerate() {
                    self.in_bufs[ch * FRAME_SIZE + i] = samp.to_sample::<f32>() * 32768.0;
                }
            }
            self.states.process(&mut self.out_bufs[..], &self.in_bufs[..], &mut self.vad[..], 1);
            !self.input.is_exhausted()
        }
    }

    impl<'model, S: Signal> Signal for DenoiseSignal<'model, S> {
        type Frame = <<S as Signal>::Frame as Frame>::Float;

        fn is_exhausted(&self) -> bool {
            self.out_idx >= FRAME_SIZE
        }
        fn next(&mut self) -> Self::Frame {
            if self.out_idx >= FRAME_SIZE {
                return Self::Frame::EQUILIBRIUM;
            }
            let idx = self.out_idx;
            self.out_idx += 1;
            let ret = Frame::from_fn(|ch| {
                let samp = (self.out_bufs[ch * FRAME_SIZE + idx] / 32768.0).clamp(-1.0, 1.0);
                samp.to_sample()
            });
            if self.out_idx >= FRAME_SIZE {
                if self.refill_out_bufs() {
                    self.out_idx = 0;
                }
            }
            ret
        }
    }
    #[allow(dead_code)]
    fn _frame_size_agrees() {
        let _: [(); FRAME_SIZE] = [(); DenoiseState::FRAME_SIZE];
    }
}
#[cfg(feature = "dasp")]
pub use signal::DenoiseSignal;

/// Page-locked host memory (`nnn_host_alloc`): slices of it cross the bus by DMA in `BatchDenoiser::process`, uploads and
/// downloads at the same time; ordinary slices work too, through the runtime's staging copies.
pub struct PinnedBuf {
    ptr: *mut c_float,
    len: usize,
}

impl PinnedBuf {
    pub fn new(len: usize) -> Option<PinnedBuf> {
        let ptr = unsafe { nnn_host_alloc(len * std::mem::size_of::<c_float>()) } as *mut c_float;
        if ptr.is_null() {
            None
        } else {
            unsafe { std::ptr::write_bytes(ptr, 0, len) };
            Some(PinnedBuf { ptr, len })
        }
    }
}

impl std::ops::Deref for PinnedBuf {
    type Target = [f32];
    fn deref(&self) -> &[f32] {
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl std::ops::DerefMut for PinnedBuf {
    fn deref_mut(&mut self) -> &mut [f32] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { nnn_host_free(self.ptr as *mut c_void) }
    }
}

unsafe impl Send for PinnedBuf {}

/// All the GPUs of a node behind one object (include/nnn_node.h): the `states` vector of the reference's hosts
/// (src/nnnoiseless.rs:305-320) cut into contiguous shards, one batch and one host thread per device, no data between devices.
pub struct NodeDenoiser {
    raw: *mut RawNode,
    n: usize,
}
unsafe impl Send for NodeDenoiser {}
unsafe impl Sync for NodeDenoiser {}

impl NodeDenoiser {
    pub fn new(n_streams: usize, devices: &[i32], model: Option<&RnnModel>) -> Option<NodeDenoiser> {
        let m = model.map_or(std::ptr::null(), |m| m.0 as *const RawModel);
        let raw = unsafe { nnn_node_create(m, n_streams as c_int, devices.as_ptr(), devices.len() as c_int, std::ptr::null()) };
        if raw.is_null() {
            None
        } else {
            Some(NodeDenoiser { raw, n: n_streams })
        }
    }
    /// `input`/`output`: `[n_streams][n_frames][480]`; `vad`: `[n_frames][n_streams]`; every device works on its share at once.
    pub fn process(&mut self, output: &mut [f32], input: &[f32], vad: &mut [f32], n_frames: usize) {
        assert_eq!(input.len(), self.n * n_frames * DenoiseState::FRAME_SIZE);
        assert_eq!(output.len(), input.len());
        assert_eq!(vad.len(), self.n * n_frames);
        let rc = unsafe {
            nnn_node_process_host(
                self.raw,
                input.as_ptr(),
                output.as_mut_ptr(),
                vad.as_mut_ptr(),
                n_frames as c_int,
                n_frames * DenoiseState::FRAME_SIZE,
                DenoiseState::FRAME_SIZE,
            )
        };
        assert_eq!(rc, 0, "nnnoiseless-mi355x: backend error");
    }
    /// Buffers resident on the shards' own devices: one pointer (and optionally one `hipStream_t`) per shard; asynchronous, see `synchronize`.
    pub unsafe fn process_device(
        &mut self,
        d_out: &[*mut f32],
        d_in: &[*const f32],
        d_vad: Option<&[*mut f32]>,
        hip_streams: Option<&[*mut c_void]>,
        n_frames: usize,
        stream_stride: usize,
        frame_stride: usize,
    ) {
        let n = nnn_node_num_shards(self.raw) as usize;
        assert!(d_in.len() == n && d_out.len() == n, "one table entry per shard");
        assert!(d_vad.map_or(true, |v| v.len() == n) && hip_streams.map_or(true, |v| v.len() == n), "one table entry per shard");
        let rc = nnn_node_process_device_streams(
            self.raw,
            d_in.as_ptr(),
            d_out.as_ptr(),
            d_vad.map_or(std::ptr::null(), |v| v.as_ptr()),
            hip_streams.map_or(std::ptr::null(), |v| v.as_ptr()),
            n as c_int,
            n_frames as c_int,
            stream_stride,
            frame_stride,
        );
        assert_eq!(rc, 0, "nnnoiseless-mi355x: backend error");
    }
    pub fn synchronize(&mut self) {
        assert_eq!(unsafe { nnn_node_synchronize(self.raw) }, 0, "nnnoiseless-mi355x: backend error");
    }
    /// The CPUs shard `i`'s host thread is pinned to (its device's local CPUs); empty when not pinned.
    pub fn shard_cpus(&self, i: usize) -> String {
        unsafe { std::ffi::CStr::from_ptr(nnn_node_shard_cpus(self.raw, i as c_int)).to_string_lossy().into_owned() }
    }
    /// Also clears the failed state a call leaves behind when one shard fails (the others have advanced).
    pub fn reset(&mut self) {
        unsafe { nnn_node_reset(self.raw) };
    }
}
impl Drop for NodeDenoiser {
    fn drop(&mut self) {
        unsafe { nnn_node_destroy(self.raw) }
    }
}
