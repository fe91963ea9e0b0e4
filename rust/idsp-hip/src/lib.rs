//! `dsp_process` trait surface over `libidsp_hip.so` — the MI355X bulk engine for `idsp`'s per-sample filter
//! hot path.  One launch replaces the reference's "block loop x process body x N lanes" triple loop
//! (`dsp-process/src/process.rs:122-141` driven by `Lanes`, `dsp-process/src/compose.rs:468-513`).
//!
//! The reference's lane count is a const generic with `[S; N]` state on the stack (`compose.rs:468,478`);
//! 65536 lanes of state do not belong on a stack, so the lane count here is a run-time value and the states live
//! in HBM ([`GpuState`]).  Everything else keeps the reference's shape:
//!
//! ```ignore
//! use dsp_process::{LaneMajor, Split, ViewProcess};
//! use idsp::iir::{Biquad, DirectForm1};
//! use idsp_hip::{DevBuf, DevView, DevViewMut, GpuLanes, GpuState};
//!
//! // reference:  Split::new(biquad, DirectForm1::default()).lanes::<N>()
//! let mut p = Split::new(GpuLanes::new(biquad), GpuState::<DirectForm1<i32>>::new(lanes)?);
//! // reference:  p.process_view(View::<_, LaneMajor, N>::from_flat(x, frames), ViewMut::from_flat(y, frames))
//! p.process_view(DevView::<_, LaneMajor>::from_flat(&x, lanes, frames), DevViewMut::from_flat(&mut y, lanes, frames));
//! ```
//!
//! Written against `include/idsp_hip.h`; NOT compiled in the repository's build image (no Rust toolchain there).
//! tests/test_rust_shim.py checks every `sys::idsp_*` used below against the generated `idsp-hip-sys` block.

use core::ffi::{c_int, c_void};
use core::marker::PhantomData;

use dsp_fixedpoint::Q32;
use dsp_process::{FrameMajor, LaneMajor, SplitViewInplace, SplitViewProcess};
use idsp::iir::{Biquad, BiquadClamp, DirectForm1, DirectForm1Dither, DirectForm1Wide, DirectForm2Transposed};
use idsp::{Complex, Lockin, Lowpass, LowpassState};

pub use idsp_hip_sys as sys;

// ---------------------------------------------------------------------------------------------- errors
/// A negative `idsp_status` with the library's thread-local description.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl core::fmt::Display for Error {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "idsp status {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(rc: c_int) -> Result<(), Error> {
    if rc == 0 {
        return Ok(());
    }
    if rc > 0 {
        // a positive status is a callback's stop code returned verbatim by idsp_multi_for_each: not a success
        return Err(Error { code: rc, message: String::from("stopped early by the callback (idsp_multi_last_block() tells where)") });
    }
    // SAFETY: idsp_last_error returns a pointer to a NUL-terminated thread-local buffer owned by the library.
    let message = unsafe { core::ffi::CStr::from_ptr(sys::idsp_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// A `hipStream_t` (NULL = the default stream).  Calls are asynchronous on it.
#[derive(Clone, Copy, Debug)]
pub struct Stream(pub *mut c_void);
impl Default for Stream {
    fn default() -> Self {
        Self(core::ptr::null_mut())
    }
}
impl Stream {
    pub fn sync(self) -> Result<(), Error> {
        // SAFETY: plain FFI call on a stream handle the caller vouches for.
        check(unsafe { sys::idsp_stream_sync(self.0) })
    }
}

/// Wait for ALL work of the current device, whatever stream it was launched on (`hipDeviceSynchronize`).
/// Launches may run on caller-chosen streams (`GpuLanes::on`, the non-blocking `MultiGpu` streams), which a
/// NULL-stream copy does not wait for; every host read-back and every buffer replacement below goes through
/// this first, so safe code can never observe a result before its kernel has finished.
pub fn device_sync() -> Result<(), Error> {
    // SAFETY: plain FFI call without arguments.
    check(unsafe { sys::idsp_device_sync() })
}

/// The library must be at least the ABI revision this binding was generated from (`idsp-hip-sys` links symbols that
/// revision 1 does not export).  Checked once, on the first allocation.
fn ensure_abi() -> Result<(), Error> {
    use std::sync::OnceLock;
    static OK: OnceLock<bool> = OnceLock::new();
    // SAFETY: plain FFI call without arguments.
    let ok = *OK.get_or_init(|| unsafe { sys::idsp_version() } >= sys::IDSP_ABI_VERSION as c_int);
    if ok {
        Ok(())
    } else {
        Err(Error { code: sys::IDSP_EINVAL as i32, message: format!("libidsp_hip.so is older than ABI revision {} this binding was generated from", sys::IDSP_ABI_VERSION) })
    }
}

// -------------------------------------------------------------------------------------- device buffers
/// Owned device memory holding `len` values of `T` (allocated through `idsp_device_alloc`).
pub struct DevBuf<T> {
    ptr: *mut T,
    len: usize,
}

impl<T: Copy> DevBuf<T> {
    /// Zero-filled buffer (a zero-filled state is the reference's `Default::default()`).
    pub fn zeroed(len: usize) -> Result<Self, Error> {
        ensure_abi()?;
        let mut raw: *mut c_void = core::ptr::null_mut();
        let bytes = len * core::mem::size_of::<T>();
        // SAFETY: `raw` is a valid out-pointer; the library allocates with hipMalloc.
        check(unsafe { sys::idsp_device_alloc(&mut raw, bytes.max(1)) })?;
        let buf = Self { ptr: raw.cast(), len };
        // SAFETY: the allocation covers `bytes`.
        check(unsafe { sys::idsp_device_memset(raw, 0, bytes, core::ptr::null_mut()) })?;
        Stream::default().sync()?;
        Ok(buf)
    }

    pub fn from_host(src: &[T]) -> Result<Self, Error> {
        let buf = Self::zeroed(src.len())?;
        // SAFETY: both ranges hold src.len() values of T.
        check(unsafe {
            sys::idsp_device_h2d(buf.ptr.cast(), src.as_ptr().cast(), core::mem::size_of_val(src), core::ptr::null_mut())
        })?;
        Stream::default().sync()?;
        Ok(buf)
    }

    pub fn to_host(&self, dst: &mut [T]) -> Result<(), Error> {
        assert_eq!(dst.len(), self.len);
        device_sync()?; // the producer may have run on any stream
        // SAFETY: both ranges hold self.len values of T.
        check(unsafe {
            sys::idsp_device_d2h(dst.as_mut_ptr().cast(), self.ptr.cast_const().cast(), core::mem::size_of_val(dst), core::ptr::null_mut())
        })?;
        Stream::default().sync()
    }

    #[must_use]
    pub fn len(&self) -> usize {
        self.len
    }
    #[must_use]
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    #[must_use]
    pub fn as_ptr(&self) -> *const T {
        self.ptr
    }
    #[must_use]
    pub fn as_mut_ptr(&mut self) -> *mut T {
        self.ptr
    }
}

impl<T> Drop for DevBuf<T> {
    fn drop(&mut self) {
        // SAFETY: allocated by idsp_device_alloc, freed once.
        unsafe { sys::idsp_device_free(self.ptr.cast()) };
    }
}

/// Layout markers of `dsp_process::view` mapped to the ABI's `layout` argument.
pub trait Layout {
    const ID: c_int;
}
impl Layout for FrameMajor {
    const ID: c_int = sys::IDSP_FRAME_MAJOR; // dsp-process/src/view.rs:10
}
impl Layout for LaneMajor {
    const ID: c_int = sys::IDSP_LANE_MAJOR; // dsp-process/src/view.rs:17
}

/// Device-side twin of `dsp_process::View<'a, T, Layout, L>` with a run-time lane count.
#[derive(Clone, Copy)]
pub struct DevView<'a, T, L> {
    ptr: *const T,
    lanes: usize,
    frames: usize,
    _p: PhantomData<(&'a T, L)>,
}

/// Device-side twin of `dsp_process::ViewMut`.
pub struct DevViewMut<'a, T, L> {
    ptr: *mut T,
    lanes: usize,
    frames: usize,
    _p: PhantomData<(&'a mut T, L)>,
}

impl<'a, T: Copy, L: Layout> DevView<'a, T, L> {
    /// `flat.len()` must equal `lanes * frames` (`View::from_flat`, dsp-process/src/view.rs:181-188).
    #[must_use]
    pub fn from_flat(flat: &'a DevBuf<T>, lanes: usize, frames: usize) -> Self {
        assert_eq!(flat.len(), lanes * frames);
        Self { ptr: flat.as_ptr(), lanes, frames, _p: PhantomData }
    }
    #[must_use]
    pub fn frames(&self) -> usize {
        self.frames
    }
    #[must_use]
    pub fn lanes(&self) -> usize {
        self.lanes
    }
}

impl<'a, T: Copy, L: Layout> DevViewMut<'a, T, L> {
    #[must_use]
    pub fn from_flat(flat: &'a mut DevBuf<T>, lanes: usize, frames: usize) -> Self {
        assert_eq!(flat.len(), lanes * frames);
        Self { ptr: flat.as_mut_ptr(), lanes, frames, _p: PhantomData }
    }
    #[must_use]
    pub fn frames(&self) -> usize {
        self.frames
    }
    #[must_use]
    pub fn lanes(&self) -> usize {
        self.lanes
    }
}

// ---------------------------------------------------------------------------------------------- state
/// How a reference state type maps onto the ABI's per-lane record of 32-bit words (include/idsp_hip.h; the reference
/// structs carry no `repr(C)`, src/iir/biquad.rs:258-269, so they are marshalled field by field).
pub trait StateRecord: Sized {
    const WORDS: usize;
    fn to_words(&self, w: &mut [u32]);
    fn from_words(w: &[u32]) -> Self;
}

fn lo_hi(v: i64) -> (u32, u32) {
    (v as u32, (v >> 32) as u32)
}
fn from_lo_hi(lo: u32, hi: u32) -> i64 {
    ((hi as i64) << 32) | lo as i64
}

/// `DirectForm1<i32>`: { x0, x1, y0, y1 }
impl StateRecord for DirectForm1<i32> {
    const WORDS: usize = 4;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.x[0] as u32;
        w[1] = self.x[1] as u32;
        w[2] = self.y[0][0] as u32;
        w[3] = self.y[0][1] as u32;
    }
    fn from_words(w: &[u32]) -> Self {
        Self { x: [w[0] as i32, w[1] as i32], y: [[w[2] as i32, w[3] as i32]] }
    }
}
/// `DirectForm1<f32>`: the same four values as IEEE bits
impl StateRecord for DirectForm1<f32> {
    const WORDS: usize = 4;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.x[0].to_bits();
        w[1] = self.x[1].to_bits();
        w[2] = self.y[0][0].to_bits();
        w[3] = self.y[0][1].to_bits();
    }
    fn from_words(w: &[u32]) -> Self {
        Self { x: [f32::from_bits(w[0]), f32::from_bits(w[1])], y: [[f32::from_bits(w[2]), f32::from_bits(w[3])]] }
    }
}
/// `DirectForm2Transposed<f32>` = `DirectForm<f32, 0, 2>`: { s0, s1 } (src/iir/biquad.rs:407)
impl StateRecord for DirectForm2Transposed<f32> {
    const WORDS: usize = 2;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.x[0].to_bits();
        w[1] = self.x[1].to_bits();
    }
    fn from_words(w: &[u32]) -> Self {
        Self { x: [f32::from_bits(w[0]), f32::from_bits(w[1])], y: [] }
    }
}
/// `DirectForm1Dither`: { x0, x1, y0, y1, e } (src/iir/biquad.rs:484-491)
impl StateRecord for DirectForm1Dither {
    const WORDS: usize = 5;
    fn to_words(&self, w: &mut [u32]) {
        self.xy.to_words(&mut w[..4]);
        w[4] = self.e;
    }
    fn from_words(w: &[u32]) -> Self {
        Self { xy: DirectForm1::<i32>::from_words(&w[..4]), e: w[4] }
    }
}
/// `DirectForm1Wide`: { x0, x1, y0.lo, y0.hi, y1.lo, y1.hi } (src/iir/biquad.rs:445-454)
impl StateRecord for DirectForm1Wide {
    const WORDS: usize = 6;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.x[0] as u32;
        w[1] = self.x[1] as u32;
        (w[2], w[3]) = lo_hi(self.y[0]);
        (w[4], w[5]) = lo_hi(self.y[1]);
    }
    fn from_words(w: &[u32]) -> Self {
        Self { x: [w[0] as i32, w[1] as i32], y: [from_lo_hi(w[2], w[3]), from_lo_hi(w[4], w[5])] }
    }
}

/// Lock-in state: the phase accumulator `{ accu.state, accu.step }` followed by `[[LowpassState<N>; K]; 2]`
/// (index 0 = I, 1 = Q), every i64 as lo, hi — include/idsp_hip.h, `idsp_lockin_state_words`.
#[derive(Clone, Debug)]
pub struct LockinState<const N: usize, const K: usize> {
    /// `Accu<Wrapping<i32>>` (src/accu.rs:16-41): current phase and phase increment per sample
    pub phase: i32,
    pub step: i32,
    pub iq: [[LowpassState<N>; K]; 2],
}
impl<const N: usize, const K: usize> StateRecord for LockinState<N, K> {
    const WORDS: usize = 2 + 2 * K * N * 2;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.phase as u32;
        w[1] = self.step as u32;
        for (q, arm) in self.iq.iter().enumerate() {
            for (c, st) in arm.iter().enumerate() {
                for j in 0..N {
                    let i = 2 + ((q * K + c) * N + j) * 2;
                    (w[i], w[i + 1]) = lo_hi(st.0[j]);
                }
            }
        }
    }
    fn from_words(w: &[u32]) -> Self {
        let iq = core::array::from_fn(|q| {
            core::array::from_fn(|c| {
                LowpassState(core::array::from_fn(|j| {
                    let i = 2 + ((q * K + c) * N + j) * 2;
                    from_lo_hi(w[i], w[i + 1])
                }))
            })
        });
        Self { phase: w[0] as i32, step: w[1] as i32, iq }
    }
}

/// `[S; lanes]` in HBM: `S::WORDS` planes of `lanes` words, word `w` of lane `l` at `w * lanes + l`, so that state
/// load/store is coalesced.  `new` = every lane `S::default()` (all-zero words).
pub struct GpuState<S> {
    words: DevBuf<u32>,
    lanes: usize,
    sections: usize,
    _s: PhantomData<S>,
}

impl<S: StateRecord> GpuState<S> {
    pub fn new(lanes: usize) -> Result<Self, Error> {
        Self::with_sections(lanes, 1)
    }
    /// State of `sections` serial sections per lane (slice composition `[C] x [S]`, compose.rs:43-77).
    pub fn with_sections(lanes: usize, sections: usize) -> Result<Self, Error> {
        Ok(Self { words: DevBuf::zeroed(S::WORDS * sections * lanes)?, lanes, sections, _s: PhantomData })
    }
    /// Upload host states: `states[l * sections + k]` = section k of lane l.
    pub fn upload(&mut self, states: &[S]) -> Result<(), Error> {
        assert_eq!(states.len(), self.lanes * self.sections);
        let mut planes = vec![0u32; S::WORDS * self.sections * self.lanes];
        let mut rec = vec![0u32; S::WORDS];
        for l in 0..self.lanes {
            for k in 0..self.sections {
                states[l * self.sections + k].to_words(&mut rec);
                for (w, v) in rec.iter().enumerate() {
                    planes[(k * S::WORDS + w) * self.lanes + l] = *v;
                }
            }
        }
        device_sync()?; // a launch on another stream may still be using the buffer that is replaced here
        self.words = DevBuf::from_host(&planes)?;
        Ok(())
    }
    pub fn download(&self) -> Result<Vec<S>, Error> {
        let mut planes = vec![0u32; S::WORDS * self.sections * self.lanes];
        self.words.to_host(&mut planes)?;
        let mut rec = vec![0u32; S::WORDS];
        let mut out = Vec::with_capacity(self.lanes * self.sections);
        for l in 0..self.lanes {
            for k in 0..self.sections {
                for (w, v) in rec.iter_mut().enumerate() {
                    *v = planes[(k * S::WORDS + w) * self.lanes + l];
                }
                out.push(S::from_words(&rec));
            }
        }
        Ok(out)
    }
    #[must_use]
    pub fn lanes(&self) -> usize {
        self.lanes
    }
}

// --------------------------------------------------------------------------------------------- kernels
/// A (configuration, state) pairing the engine has an entry point for — the GPU-side meaning of the reference's
/// `impl SplitProcess<X, Y, S> for C`.  A pairing without an impl is a compile error, like a missing trait impl in the
/// reference.
pub trait GpuKernel<S> {
    type In: Copy;
    type Out: Copy;
    /// Launch over `lanes` lanes x `frames` samples; `y == x` is the reference's `inplace`.
    ///
    /// # Safety
    /// `state`, `x`, `y` must be device pointers covering the shapes implied by `lanes`, `frames`, `layout`.
    unsafe fn launch(&self, state: *mut c_void, x: *const Self::In, y: *mut Self::Out, lanes: usize, frames: usize, layout: c_int, stream: Stream)
    -> c_int;
}

fn ba_bits<const F: i8>(b: &Biquad<Q32<F>>) -> [i32; 5] {
    // `Q` is `repr(transparent)` over its integer (dsp-fixedpoint/src/lib.rs:147)
    core::array::from_fn(|i| b.ba[i].into_bits())
}
fn cfg_i32<const F: i8>(b: &Biquad<Q32<F>>) -> sys::IdspBiquadI32 {
    sys::IdspBiquadI32 { ba: ba_bits(b), frac: F as i32 }
}
fn cfg_clamp_i32<const F: i8>(b: &BiquadClamp<Q32<F>, i32>) -> sys::IdspBiquadClampI32 {
    sys::IdspBiquadClampI32 { ba: ba_bits(&b.coeff), frac: F as i32, u: b.u, min: b.min, max: b.max }
}

macro_rules! biquad_i32_kernel {
    ($cfg:ty, $state:ty, $conv:ident, $entry:ident) => {
        impl<const F: i8> GpuKernel<$state> for $cfg {
            type In = i32;
            type Out = i32;
            unsafe fn launch(&self, state: *mut c_void, x: *const i32, y: *mut i32, lanes: usize, frames: usize, layout: c_int, stream: Stream) -> c_int {
                let c = $conv(self);
                // SAFETY: forwarded from the caller's contract; `c` outlives the call (the library copies it into kernel arguments).
                unsafe { sys::$entry(&c, 1, state, x, y, lanes, frames, layout, stream.0) }
            }
        }
    };
}
// src/iir/biquad.rs:366-383, 394-404, 511-538, 456-480
biquad_i32_kernel!(Biquad<Q32<F>>, DirectForm1<i32>, cfg_i32, idsp_biquad_i32_df1);
biquad_i32_kernel!(BiquadClamp<Q32<F>, i32>, DirectForm1<i32>, cfg_clamp_i32, idsp_biquad_i32_df1_clamp);
biquad_i32_kernel!(Biquad<Q32<F>>, DirectForm1Dither, cfg_i32, idsp_biquad_i32_dither);
biquad_i32_kernel!(BiquadClamp<Q32<F>, i32>, DirectForm1Dither, cfg_clamp_i32, idsp_biquad_i32_dither_clamp);
biquad_i32_kernel!(Biquad<Q32<F>>, DirectForm1Wide, cfg_i32, idsp_biquad_i32_wide);
biquad_i32_kernel!(BiquadClamp<Q32<F>, i32>, DirectForm1Wide, cfg_clamp_i32, idsp_biquad_i32_wide_clamp);

macro_rules! biquad_f32_kernel {
    ($cfg:ty, $state:ty, $conv:expr, $entry:ident) => {
        impl GpuKernel<$state> for $cfg {
            type In = f32;
            type Out = f32;
            unsafe fn launch(&self, state: *mut c_void, x: *const f32, y: *mut f32, lanes: usize, frames: usize, layout: c_int, stream: Stream) -> c_int {
                let c = $conv(self);
                // SAFETY: see biquad_i32_kernel.
                unsafe { sys::$entry(&c, 1, state, x, y, lanes, frames, layout, stream.0) }
            }
        }
    };
}
// src/iir/biquad.rs:366-383 (C = T = f32), 418-440
biquad_f32_kernel!(Biquad<f32>, DirectForm1<f32>, |b: &Biquad<f32>| sys::IdspBiquadF32 { ba: b.ba }, idsp_biquad_f32_df1);
biquad_f32_kernel!(Biquad<f32>, DirectForm2Transposed<f32>, |b: &Biquad<f32>| sys::IdspBiquadF32 { ba: b.ba }, idsp_biquad_f32_df2t);
biquad_f32_kernel!(
    BiquadClamp<f32, f32>,
    DirectForm1<f32>,
    |b: &BiquadClamp<f32, f32>| sys::IdspBiquadClampF32 { ba: b.coeff.ba, u: b.u, min: b.min, max: b.max },
    idsp_biquad_f32_df1_clamp
);
biquad_f32_kernel!(
    BiquadClamp<f32, f32>,
    DirectForm2Transposed<f32>,
    |b: &BiquadClamp<f32, f32>| sys::IdspBiquadClampF32 { ba: b.coeff.ba, u: b.u, min: b.min, max: b.max },
    idsp_biquad_f32_df2t_clamp
);

/// `Lockin<[Lowpass<N>; K]>` fed by a per-lane phase accumulator (src/lockin.rs:30-39, src/lowpass.rs:47-78):
/// input `i32`, output `Complex<i32>` = `[re, im]` (`repr(transparent)`, src/complex.rs:15-20).
impl<const N: usize, const K: usize> GpuKernel<LockinState<N, K>> for Lockin<[Lowpass<N>; K]> {
    type In = i32;
    type Out = Complex<i32>;
    unsafe fn launch(&self, state: *mut c_void, x: *const i32, y: *mut Complex<i32>, lanes: usize, frames: usize, layout: c_int, stream: Stream) -> c_int {
        let mut k = [[0i32; 2]; sys::IDSP_LOCKIN_MAX_CASCADE];
        for (c, lp) in self.0.iter().enumerate() {
            k[c][..N].copy_from_slice(&lp.0);
        }
        let cfg = sys::IdspLockinI32 { order: N as i32, cascade: K as i32, k };
        // SAFETY: Complex<i32> is repr(transparent) over [i32; 2]; rest as above.
        unsafe { sys::idsp_lockin_i32_process(&cfg, state, x, y.cast(), lanes, frames, layout, stream.0) }
    }
}

/// Lock-in state with biquad arms: `{ accu.state, accu.step }` followed by `[[DirectForm1<i32>; NS]; 2]` (index 0 = I,
/// 1 = Q), four words each — include/idsp_hip.h, `idsp_lockin_biquad_state_words(NS, 1)`.
#[derive(Clone, Debug)]
pub struct LockinBiquadState<const NS: usize> {
    pub phase: i32,
    pub step: i32,
    pub iq: [[DirectForm1<i32>; NS]; 2],
}
impl<const NS: usize> StateRecord for LockinBiquadState<NS> {
    const WORDS: usize = 2 + 8 * NS;
    fn to_words(&self, w: &mut [u32]) {
        w[0] = self.phase as u32;
        w[1] = self.step as u32;
        for (q, arm) in self.iq.iter().enumerate() {
            for (k, st) in arm.iter().enumerate() {
                st.to_words(&mut w[2 + (q * NS + k) * 4..][..4]);
            }
        }
    }
    fn from_words(w: &[u32]) -> Self {
        let iq = core::array::from_fn(|q| core::array::from_fn(|k| DirectForm1::<i32>::from_words(&w[2 + (q * NS + k) * 4..][..4])));
        Self { phase: w[0] as i32, step: w[1] as i32, iq }
    }
}

/// `Lockin<[Biquad<Q32<F>>; NS]>` (any arm filter `C: SplitProcess<i32, i32, S>`, src/lockin.rs:16-39) fed by a per-lane
/// phase accumulator: the same `NS` DF1 sections on I and on Q.  `Lockin<Biquad<Q32<F>>>` is `NS = 1`.
impl<const F: i8, const NS: usize> GpuKernel<LockinBiquadState<NS>> for Lockin<[Biquad<Q32<F>>; NS]> {
    type In = i32;
    type Out = Complex<i32>;
    unsafe fn launch(&self, state: *mut c_void, x: *const i32, y: *mut Complex<i32>, lanes: usize, frames: usize, layout: c_int, stream: Stream) -> c_int {
        let sec: [sys::IdspBiquadI32; NS] = core::array::from_fn(|k| cfg_i32(&self.0[k]));
        // SAFETY: Complex<i32> is repr(transparent) over [i32; 2]; `sec` outlives the call (the library copies it into the launch).
        unsafe { sys::idsp_lockin_i32_biquad_process(sec.as_ptr(), NS, state, x, y.cast(), lanes, frames, layout, stream.0) }
    }
}

/// External-LO form `(x, Complex<U>) -> Complex<X>` (src/lockin.rs:17-27): two input buffers, so it is a method rather
/// than a `GpuKernel`.  `Lockin<[Biquad<f32>; NS]>` with lo = (cos, -sin) is the graph of examples/ddc_lockin.rs:35-42.
pub trait GpuLockinLo<S> {
    type X: Copy;
    type U: Copy;
    /// # Safety
    /// `state` holds `lanes` records of `[S; 2]` in the library's plane layout; x, lo and y are device buffers of
    /// lanes * frames (x) and lanes * frames pairs (lo, y) in `layout`.
    unsafe fn launch_lo(&self, state: *mut c_void, x: *const Self::X, lo: *const Complex<Self::U>, y: *mut Complex<Self::X>, lanes: usize,
                        frames: usize, layout: c_int, stream: Stream) -> c_int;
}
impl<const N: usize, const K: usize> GpuLockinLo<[LowpassState<N>; K]> for Lockin<[Lowpass<N>; K]> {
    type X = i32;
    type U = Q32<32>;
    unsafe fn launch_lo(&self, state: *mut c_void, x: *const i32, lo: *const Complex<Q32<32>>, y: *mut Complex<i32>, lanes: usize, frames: usize,
                        layout: c_int, stream: Stream) -> c_int {
        let mut k = [[0i32; 2]; sys::IDSP_LOCKIN_MAX_CASCADE];
        for (c, lp) in self.0.iter().enumerate() {
            k[c][..N].copy_from_slice(&lp.0);
        }
        let cfg = sys::IdspLockinI32 { order: N as i32, cascade: K as i32, k };
        // SAFETY: Q32 and Complex are repr(transparent): Complex<Q32<32>> is [i32; 2].
        unsafe { sys::idsp_lockin_i32_lo_process(&cfg, state, x, lo.cast(), y.cast(), lanes, frames, layout, stream.0) }
    }
}
impl<const F: i8, const NS: usize> GpuLockinLo<[DirectForm1<i32>; NS]> for Lockin<[Biquad<Q32<F>>; NS]> {
    type X = i32;
    type U = Q32<32>;
    unsafe fn launch_lo(&self, state: *mut c_void, x: *const i32, lo: *const Complex<Q32<32>>, y: *mut Complex<i32>, lanes: usize, frames: usize,
                        layout: c_int, stream: Stream) -> c_int {
        let sec: [sys::IdspBiquadI32; NS] = core::array::from_fn(|k| cfg_i32(&self.0[k]));
        // SAFETY: as above.
        unsafe { sys::idsp_lockin_i32_biquad_lo_process(sec.as_ptr(), NS, state, x, lo.cast(), y.cast(), lanes, frames, layout, stream.0) }
    }
}
impl<const NS: usize> GpuLockinLo<[DirectForm1<f32>; NS]> for Lockin<[Biquad<f32>; NS]> {
    type X = f32;
    type U = f32;
    unsafe fn launch_lo(&self, state: *mut c_void, x: *const f32, lo: *const Complex<f32>, y: *mut Complex<f32>, lanes: usize, frames: usize,
                        layout: c_int, stream: Stream) -> c_int {
        let sec: [sys::IdspBiquadF32; NS] = core::array::from_fn(|k| sys::IdspBiquadF32 { ba: self.0[k].ba });
        // SAFETY: as above.
        unsafe { sys::idsp_lockin_f32_biquad_lo_process(sec.as_ptr(), NS, state, x, lo.cast(), y.cast(), lanes, frames, layout, stream.0) }
    }
}

// ----------------------------------------------------------------------------------------------- lanes
/// GPU-side `Lanes<C>` (dsp-process/src/compose.rs:449-513): one configuration shared by all lanes.
#[derive(Clone, Copy, Debug, Default)]
pub struct GpuLanes<C> {
    pub config: C,
    pub stream: Stream,
}

impl<C> GpuLanes<C> {
    #[must_use]
    pub const fn new(config: C) -> Self {
        Self { config, stream: Stream(core::ptr::null_mut()) }
    }
    #[must_use]
    pub fn on(mut self, stream: Stream) -> Self {
        self.stream = stream;
        self
    }

    /// Fallible form of `process_view`: shape violations that are `debug_assert!` in the reference
    /// (dsp-process/src/process.rs:42-45) come back as `Error { code: IDSP_EINVAL, .. }`.
    pub fn try_process_view<S, L>(&self, state: &mut GpuState<S>, x: DevView<'_, C::In, L>, y: DevViewMut<'_, C::Out, L>) -> Result<(), Error>
    where
        C: GpuKernel<S>,
        S: StateRecord,
        L: Layout,
    {
        assert_eq!(x.frames, y.frames); // process.rs:42-45
        assert_eq!(x.lanes, y.lanes);
        assert_eq!(x.lanes, state.lanes);
        // SAFETY: the views were built from DevBufs of lanes * frames values; the state buffer holds S::WORDS planes.
        check(unsafe { self.config.launch(state.words.as_mut_ptr().cast(), x.ptr, y.ptr, x.lanes, x.frames, L::ID, self.stream) })
    }

    pub fn try_inplace_view<S, L>(&self, state: &mut GpuState<S>, xy: DevViewMut<'_, C::In, L>) -> Result<(), Error>
    where
        C: GpuKernel<S, Out = <C as GpuKernel<S>>::In>,
        S: StateRecord,
        L: Layout,
    {
        assert_eq!(xy.lanes, state.lanes);
        // SAFETY: as above with y == x (the ABI allows exact aliasing for same-rate operators).
        check(unsafe { self.config.launch(state.words.as_mut_ptr().cast(), xy.ptr.cast_const(), xy.ptr, xy.lanes, xy.frames, L::ID, self.stream) })
    }
}

/// `SplitViewProcess` (dsp-process/src/view.rs:245-249) for device views in either layout: LaneMajor is
/// `Lanes::process_view` (compose.rs:478-494), FrameMajor is `SplitProcess::block` over `[[X; lanes]]`
/// (process.rs:122-127 on `Lanes::process`, compose.rs:468-476).  Like the reference's, it cannot report errors:
/// a failed launch panics with the library's message; use `try_process_view` to handle it.
impl<'a, 'b, C, S, L> SplitViewProcess<DevView<'a, C::In, L>, DevViewMut<'b, C::Out, L>, GpuState<S>> for GpuLanes<C>
where
    C: GpuKernel<S>,
    S: StateRecord,
    L: Layout,
{
    fn process_view(&self, state: &mut GpuState<S>, x: DevView<'a, C::In, L>, y: DevViewMut<'b, C::Out, L>) {
        self.try_process_view(state, x, y).unwrap_or_else(|e| panic!("{e}"));
    }
}

/// `SplitViewInplace` (dsp-process/src/view.rs:251-255; `Lanes` impl compose.rs:504-513).
impl<'a, C, S, L> SplitViewInplace<DevViewMut<'a, C::In, L>, GpuState<S>> for GpuLanes<C>
where
    C: GpuKernel<S, Out = <C as GpuKernel<S>>::In>,
    S: StateRecord,
    L: Layout,
{
    fn inplace_view(&self, state: &mut GpuState<S>, xy: DevViewMut<'a, C::In, L>) {
        self.try_inplace_view(state, xy).unwrap_or_else(|e| panic!("{e}"));
    }
}

// ------------------------------------------------------------------------------- half-band cascades
/// `HBF_DEC_CASCADE` / `HBF_INT_CASCADE` restricted to a 2^stages rate change (src/hbf.rs:385-421, 476-512) with the
/// built-in tap sets (0 = `HBF_TAPS`, 1 = `HBF_TAPS_98`), over many lanes.  State = the history the reference keeps by
/// `copy_within` (src/hbf.rs:182-183, 224), zero-initialised like `HbfDec16::default()`.
pub struct GpuHbf {
    cfg: sys::IdspHbfCascadeF32,
    state: DevBuf<u32>,
    lanes: usize,
    decimate: bool,
    pub stream: Stream,
}

impl GpuHbf {
    pub fn decimator(tap_set: i32, stages: i32, lanes: usize) -> Result<Self, Error> {
        Self::build(tap_set, stages, lanes, true)
    }
    pub fn interpolator(tap_set: i32, stages: i32, lanes: usize) -> Result<Self, Error> {
        Self::build(tap_set, stages, lanes, false)
    }
    fn build(tap_set: i32, stages: i32, lanes: usize, decimate: bool) -> Result<Self, Error> {
        let mut cfg = sys::IdspHbfCascadeF32 { stages: 0, m: [0; sys::IDSP_HBF_MAX_STAGES], taps: [[0.0; sys::IDSP_HBF_MAX_TAPS]; sys::IDSP_HBF_MAX_STAGES] };
        // SAFETY: `cfg` is a valid out-record.
        check(unsafe { if decimate { sys::idsp_hbf_dec_cascade(tap_set, stages, &mut cfg) } else { sys::idsp_hbf_int_cascade(tap_set, stages, &mut cfg) } })?;
        // SAFETY: `cfg` was filled by the library.
        let words = unsafe { if decimate { sys::idsp_hbf_dec_state_words(&cfg) } else { sys::idsp_hbf_int_state_words(&cfg) } };
        Ok(Self { cfg, state: DevBuf::zeroed(words * lanes)?, lanes, decimate, stream: Stream::default() })
    }
    /// Rate change R = 2^stages.
    #[must_use]
    pub fn rate(&self) -> usize {
        1 << self.cfg.stages
    }
    /// `frames` = low-rate samples per lane; the high-rate side holds chunks `[f32; R]`
    /// (`SplitProcess<[f32; R], f32, _>` / `SplitProcess<f32, [f32; R], _>`, src/hbf.rs:156-236).
    pub fn block<L: Layout>(&mut self, x: &DevBuf<f32>, y: &mut DevBuf<f32>, frames: usize) -> Result<(), Error> {
        let (hi, lo) = if self.decimate { (x.len(), y.len()) } else { (y.len(), x.len()) };
        assert_eq!(lo, self.lanes * frames);
        assert_eq!(hi, self.lanes * frames * self.rate());
        // SAFETY: buffer sizes checked above.
        check(unsafe {
            if self.decimate {
                sys::idsp_hbf_dec_f32(&self.cfg, self.state.as_mut_ptr().cast(), x.as_ptr(), y.as_mut_ptr(), self.lanes, frames, L::ID, self.stream.0)
            } else {
                sys::idsp_hbf_int_f32(&self.cfg, self.state.as_mut_ptr().cast(), x.as_ptr(), y.as_mut_ptr(), self.lanes, frames, L::ID, self.stream.0)
            }
        })
    }
}

// ----------------------------------------------------------------------------------------- elementwise
/// `cossin(phase)` over a buffer (src/cossin.rs:14-67; the `cossin(p) -> [N, 2]` function of src/py.rs:10-28).
pub fn cossin(phase: &DevBuf<i32>, out: &mut DevBuf<Complex<i32>>, stream: Stream) -> Result<(), Error> {
    assert_eq!(phase.len(), out.len());
    // SAFETY: sizes checked; Complex<i32> is [i32; 2].
    check(unsafe { sys::idsp_cossin_i32(phase.as_ptr(), out.as_mut_ptr().cast(), phase.len(), stream.0) })
}

/// `Complex::<i32>::arg` = `atan2(im, re)` over a buffer (src/complex.rs:254-256, src/atan2.rs:66-82).
pub fn arg(z: &DevBuf<Complex<i32>>, out: &mut DevBuf<i32>, stream: Stream) -> Result<(), Error> {
    assert_eq!(z.len(), out.len());
    // SAFETY: sizes checked.
    check(unsafe { sys::idsp_atan2_i32(z.as_ptr().cast(), out.as_mut_ptr(), z.len(), stream.0) })
}

/// Number of visible devices / select one for this thread (one process per GPU is the deployment model: lanes
/// never interact, so N GPUs take N contiguous lane blocks with no data-path collective).
pub fn device_count() -> Result<usize, Error> {
    // SAFETY: plain FFI call.
    let n = unsafe { sys::idsp_device_count() };
    check(n)?;
    Ok(n as usize)
}
pub fn device_set(device: usize) -> Result<(), Error> {
    // SAFETY: plain FFI call.
    check(unsafe { sys::idsp_device_set(device as c_int) })
}

// ------------------------------------------------------------------------------ several devices, one process
/// Lane split over several devices driven from one process (`idsp_multi_*`): device g of G owns lanes
/// [g L / G, (g + 1) L / G) and there is no data-path exchange (compose.rs:468-476: lanes never interact).
pub struct MultiGpu {
    raw: *mut sys::IdspMulti,
}

impl MultiGpu {
    /// `None` = every visible device.
    pub fn new(devices: Option<&[i32]>) -> Result<Self, Error> {
        let mut raw = core::ptr::null_mut();
        let (p, n) = devices.map_or((core::ptr::null(), 0), |d| (d.as_ptr(), d.len() as c_int));
        // SAFETY: `p` covers `n` ordinals (or is NULL with n == 0); `raw` is a valid out-pointer.
        check(unsafe { sys::idsp_multi_create(p, n, &mut raw) })?;
        Ok(Self { raw })
    }
    #[must_use]
    pub fn blocks(&self) -> usize {
        // SAFETY: valid handle.
        unsafe { sys::idsp_multi_size(self.raw) as usize }
    }
    /// Lane block of block `index` for a job of `lanes` lanes.
    pub fn shard(&self, lanes: usize, index: usize) -> Result<core::ops::Range<usize>, Error> {
        let (mut lo, mut hi) = (0usize, 0usize);
        // SAFETY: valid handle and out-pointers.
        check(unsafe { sys::idsp_multi_shard(self.raw, lanes, index as c_int, &mut lo, &mut hi) })?;
        Ok(lo..hi)
    }
    /// Run `f(block index, lane block, stream)` with each block's device current; `f` issues the plain entry points
    /// (or a `GpuLanes` on that stream) for its lane block.
    pub fn for_each<F: FnMut(usize, core::ops::Range<usize>, Stream) -> Result<(), Error>>(&mut self, lanes: usize, mut f: F) -> Result<(), Error> {
        unsafe extern "C" fn tramp<F: FnMut(usize, core::ops::Range<usize>, Stream) -> Result<(), Error>>(
            user: *mut c_void, index: c_int, lo: usize, hi: usize, stream: *mut c_void,
        ) -> c_int {
            // SAFETY: `user` is the `&mut F` passed below and outlives the call.
            let f = unsafe { &mut *user.cast::<F>() };
            match f(index as usize, lo..hi, Stream(stream)) {
                Ok(()) => 0,
                Err(e) => e.code,
            }
        }
        // SAFETY: the trampoline matches idsp_shard_fn; `f` lives across the call.
        check(unsafe { sys::idsp_multi_for_each(self.raw, lanes, Some(tramp::<F>), (&mut f as *mut F).cast()) })
    }
    /// Wait for every block's stream.
    pub fn sync(&mut self) -> Result<(), Error> {
        // SAFETY: valid handle.
        check(unsafe { sys::idsp_multi_sync(self.raw) })
    }
}

impl Drop for MultiGpu {
    fn drop(&mut self) {
        // SAFETY: created by idsp_multi_create, destroyed once.
        unsafe { sys::idsp_multi_destroy(self.raw) };
    }
}
