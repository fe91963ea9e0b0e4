/*
 * idsp_hip.h — C ABI of libidsp_hip.so, the MI355X (gfx950) bulk engine for the
 * per-sample filter hot path of quartiq/idsp.
 *
 * This header is the drop-in boundary: every entry point replaces one
 * `dsp_process::{SplitProcess::block, SplitInplace::inplace,
 * SplitViewProcess::process_view}` call of the reference, executed for many
 * independent lanes at once.  Citations are `path:line` inside the reference
 * tree (idsp 0.22.0).  A Rust `extern "C"` block binding these symbols is shown
 * in INTEGRATION.md.
 *
 * Conventions common to all processing entry points
 * --------------------------------------------------
 *  - plain C types only; `x`, `y`, `state` are DEVICE pointers owned by the
 *    caller; `stream` is a `hipStream_t` passed as `void*` (NULL = default
 *    stream).  Calls are asynchronous on that stream.
 *  - the library never allocates hidden memory and keeps no mutable globals
 *    apart from the thread-local last-error string.
 *  - `lanes`  = number of independent channels (reference: const generic `N` of
 *    `Lanes<C>` / `[S; N]`, dsp-process/src/compose.rs:449-513).
 *    `frames` = samples per lane in this call (reference: slice length).
 *  - `layout` selects the two memory layouts of dsp-process/src/view.rs:10-17:
 *      IDSP_FRAME_MAJOR  `[[T; lanes]; frames]`  element (f, l) at `f*lanes + l`
 *      IDSP_LANE_MAJOR   `lanes` contiguous slices, element (f, l) at `l*frames + f`
 *    (view.rs:190-195 `lane(i) = flat[i*frames..]`).
 *  - `y == x` (in-place, reference `inplace()`) is allowed for all same-rate
 *    operators; partial overlap is not.
 *  - `state` is read at entry and written back at exit, so consecutive calls
 *    continue the stream exactly as consecutive `block()` calls do in the
 *    reference.  A zero-filled state is the reference's `Default::default()`.
 *    The reference state structs carry no `repr(C)` (src/iir/biquad.rs:258-269),
 *    so this ABI fixes its own record: a state is a sequence of 32-bit words per
 *    lane, stored WORD-PLANE-MAJOR on the device for coalescing:
 *        word w of lane l  ->  ((uint32_t*)state)[w * lanes + l]
 *    64-bit fields occupy two consecutive words (low word first).  The word
 *    lists are given with each operator below; idsp_*_state_words() return the
 *    counts.
 *  - return value: IDSP_OK (0) or a negative idsp_status.  Shape / parameter
 *    violations that are `debug_assert!`/`const assert!` in the reference
 *    (dsp-process/src/process.rs:42-45, src/iir/biquad.rs:448-450) are reported
 *    as IDSP_EINVAL; the library never aborts.  idsp_last_error() returns a
 *    thread-local description of the most recent failure.
 */
#ifndef IDSP_HIP_H
#define IDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round-1 surface.  2: + the `_pitch` twins, idsp_multi_*, idsp_last_kernel, idsp_device_sync,
 * idsp_multi_last_block; dispatch switches honoured only with IDSP_DIAG=1.  3: + `Lockin<C>` with biquad arms and
 * the external-LO forms (idsp_lockin_*_biquad*, *_lo_*), the f64 half-band / FIR entries.  4: + idsp_device_copy.
 * Versions only ADD symbols:
 * a host binding refuses a library whose idsp_version() is LOWER than the version it was generated from and accepts
 * any higher one (the rule of idsp_amd/_lib.py, __graft_entry__.py and rust/idsp-hip). */
#define IDSP_ABI_VERSION 4

typedef enum idsp_status {
    IDSP_OK = 0,
    IDSP_EINVAL = -1,  /* bad shape / parameter (reference: debug_assert / const assert) */
    IDSP_EHIP = -2,    /* HIP runtime error, text in idsp_last_error() */
    IDSP_ENODEV = -3,  /* no usable gfx950 device */
    /* builder parameter validation, `iir::Error` (src/iir/error.rs:5-16); idsp_last_error()
     * holds the reference's Display text, e.g. "parameter `frequency` is out of range" */
    IDSP_ENONFINITE = -10,   /* Error::NonFinite     */
    IDSP_ENONPOSITIVE = -11, /* Error::NonPositive   */
    IDSP_EOUTOFRANGE = -12,  /* Error::OutOfRange    */
    IDSP_EINVERTED = -13,    /* Error::InvertedRange */
    IDSP_ESIGN = -14         /* Error::SignMismatch  */
} idsp_status;

typedef enum idsp_layout {
    IDSP_FRAME_MAJOR = 0, /* dsp-process/src/view.rs:10  */
    IDSP_LANE_MAJOR = 1   /* dsp-process/src/view.rs:17  */
} idsp_layout;

/* Maximum number of serial sections accepted per call (slice composition
 * `[C] x [S]`, dsp-process/src/compose.rs:43-77). */
#define IDSP_MAX_SECTIONS 64

/* ------------------------------------------------------------------------ */
/* library / device utilities                                               */
/* ------------------------------------------------------------------------ */

/* IDSP_ABI_VERSION the library was built with. */
int idsp_version(void);
/* Thread-local text of the last error returned on this thread ("" if none). */
const char *idsp_last_error(void);
/* Diagnostic: thread-local name of the kernel (and processor instantiation) the most recent processing call
 * of this thread dispatched to, e.g. "stream_frame_major_lds<idsp::bq::Chain<idsp::bq::Df1I32<false>, 1>>";
 * "" before the first launch.  Multi-pass calls report their last pass. */
const char *idsp_last_kernel(void);
/* Number of visible HIP devices, or a negative idsp_status. */
int idsp_device_count(void);
/* Select the device used by subsequent calls of this thread. */
int idsp_device_set(int device);
/* Plain device-memory helpers so a host language without a HIP binding (the
 * Rust shim, ctypes) can own buffers.  Not used on the hot path. */
int idsp_device_alloc(void **ptr, size_t bytes);
int idsp_device_free(void *ptr);
int idsp_device_memset(void *ptr, int value, size_t bytes, void *stream);
int idsp_device_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int idsp_device_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
/* Device-to-device copy of non-overlapping buffers by a streaming kernel (16 bytes per thread, nontemporal, one contiguous
 * chunk per workgroup), asynchronous on `stream`: `copy_from_slice` for a host that owns device buffers, and the yardstick
 * bench.py prints beside the filter kernels (`copy_gbs`: what a plain copy of the same footprint reaches on this box in the
 * same run).  Any alignment and size; the bulk moves 16-byte aligned when dst and src are congruent mod 16. */
int idsp_device_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int idsp_stream_sync(void *stream);
/* Wait for ALL work of the current device, whatever stream it was launched on (hipDeviceSynchronize) — what a
 * host must call before reading results back on another stream than the one it launched on: the idsp_multi
 * streams are non-blocking, so a NULL-stream copy does NOT wait for them. */
int idsp_device_sync(void);

/* ------------------------------------------------------------------------ */
/* iir::Biquad — fixed point (src/iir/biquad.rs)                            */
/* ------------------------------------------------------------------------ */

/* `Biquad<Q32<F>>` (src/iir/biquad.rs:96-116): ba = [b0,b1,b2,a1,a2] raw Q bits
 * with a1,a2 stored exactly as used in the recurrence; frac = F, 0 <= F < 32. */
typedef struct idsp_biquad_i32 {
    int32_t ba[5];
    int32_t frac;
} idsp_biquad_i32;

/* `BiquadClamp<Q32<F>, i32>` (src/iir/biquad.rs:121-157). */
typedef struct idsp_biquad_clamp_i32 {
    int32_t ba[5];
    int32_t frac;
    int32_t u;   /* summing junction offset */
    int32_t min; /* lower limit */
    int32_t max; /* upper limit */
} idsp_biquad_clamp_i32;

/* `Biquad<f32>`. */
typedef struct idsp_biquad_f32 {
    float ba[5];
} idsp_biquad_f32;

/* `BiquadClamp<f32, f32>`. */
typedef struct idsp_biquad_clamp_f32 {
    float ba[5];
    float u;
    float min;
    float max;
} idsp_biquad_clamp_f32;

/* `Biquad<f64>` / `BiquadClamp<f64, f64>` (the generic impl of src/iir/biquad.rs:366-440 with
 * C = T = A = f64). */
typedef struct idsp_biquad_f64 {
    double ba[5];
} idsp_biquad_f64;

typedef struct idsp_biquad_clamp_f64 {
    double ba[5];
    double u;
    double min;
    double max;
} idsp_biquad_clamp_f64;

/* Coefficient ingestion, host side, once per configuration.
 * `From<[[f64;3];2]> for Biquad<C>` (src/iir/biquad.rs:545-566) followed by the
 * float -> Q conversion `round(v * 2^F)` saturating, NaN -> 0
 * (dsp-fixedpoint/src/num_traits_impl.rs:32-46).  sos = [b0,b1,b2,a0,a1,a2]
 * with the literature sign of a1/a2 (this is also the row format of
 * `sos()` in src/py.rs:49-73). */
int idsp_biquad_i32_from_sos(const double sos[6], int frac, idsp_biquad_i32 *out);
/* Same normalisation evaluated in f32 (`From<[[f32;3];2]> for Biquad<f32>`). */
int idsp_biquad_f32_from_sos(const float sos[6], idsp_biquad_f32 *out);
/* Same normalisation evaluated in f64, then each coefficient cast `as f32`
 * (`From<[f64;5]> for Biquad<f32>`, src/iir/biquad.rs:570-576). */
int idsp_biquad_f32_from_sos_f64(const double sos[6], idsp_biquad_f32 *out);
/* `From<[[f64;3];2]> for Biquad<f64>`. */
int idsp_biquad_f64_from_sos(const double sos[6], idsp_biquad_f64 *out);

/* All idsp_biquad_* / idsp_cascade_* calls process `n` serial sections
 * (1 <= n <= IDSP_MAX_SECTIONS; `cfg` points at n host-side records; n == 0 is
 * the reference's empty-slice identity, compose.rs:63-65).  Section s owns the
 * state words [s*W, (s+1)*W).
 *
 * state words per section (W):
 *   DF1        `DirectForm1<T>`        W=4  { x0, x1, y0, y1 }        biquad.rs:260-269,319
 *   DF1 dither `DirectForm1Dither`     W=5  { x0, x1, y0, y1, e }     biquad.rs:484-491
 *   DF1 wide   `DirectForm1Wide`       W=6  { x0, x1, y0.lo, y0.hi, y1.lo, y1.hi }  biquad.rs:445-454
 *   DF2T       `DirectForm2Transposed` W=2  { s0, s1 }                biquad.rs:407
 *   Cascade    `DirectForm<T, n>`      2+2n words total { x0, x1, (y0, y1) x n }    biquad.rs:260-269,324
 */

/* `Biquad<Q32<F>>` x `DirectForm1<i32>` (src/iir/biquad.rs:366-383). */
int idsp_biquad_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state,
                        const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                        int layout, void *stream);
/* `BiquadClamp<Q32<F>, i32>` x `DirectForm1<i32>` (src/iir/biquad.rs:394-404). */
int idsp_biquad_i32_df1_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                              const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                              int layout, void *stream);
/* `Biquad<Q32<F>>` x `DirectForm1Dither` (src/iir/biquad.rs:511-530). */
int idsp_biquad_i32_dither(const idsp_biquad_i32 *cfg, size_t n, void *state,
                           const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                           int layout, void *stream);
/* `BiquadClamp<Q32<F>, i32>` x `DirectForm1Dither` (src/iir/biquad.rs:532-538). */
int idsp_biquad_i32_dither_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                                 const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                                 int layout, void *stream);
/* `Biquad<Q32<F>>` x `DirectForm1Wide` (src/iir/biquad.rs:456-472). */
int idsp_biquad_i32_wide(const idsp_biquad_i32 *cfg, size_t n, void *state,
                         const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                         int layout, void *stream);
/* `BiquadClamp<Q32<F>, i32>` x `DirectForm1Wide` (src/iir/biquad.rs:474-480);
 * this is the per-section operator of `sos_clamp_wide()` in src/py.rs:76-108. */
int idsp_biquad_i32_wide_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                               const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                               int layout, void *stream);
/* `Cascade<[Biquad<Q32<F>>; n]>` x `DirectForm<i32, n>` — shared delay lines
 * (src/iir/biquad.rs:339-364).  1 <= n <= 8. */
int idsp_cascade_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state,
                         const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                         int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* iir::Biquad — f32                                                        */
/* ------------------------------------------------------------------------ */

/* `Biquad<f32>` x `DirectForm1<f32>` (src/iir/biquad.rs:366-383): every product
 * and sum individually rounded, evaluated left to right, no FMA. */
int idsp_biquad_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state,
                        const float *x, float *y, size_t lanes, size_t frames,
                        int layout, void *stream);
/* `BiquadClamp<f32>` x `DirectForm1<f32>` (src/iir/biquad.rs:394-404). */
int idsp_biquad_f32_df1_clamp(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state,
                              const float *x, float *y, size_t lanes, size_t frames,
                              int layout, void *stream);
/* `Biquad<f32>` x `DirectForm2Transposed<f32>` (src/iir/biquad.rs:418-428). */
int idsp_biquad_f32_df2t(const idsp_biquad_f32 *cfg, size_t n, void *state,
                         const float *x, float *y, size_t lanes, size_t frames,
                         int layout, void *stream);
/* `BiquadClamp<f32>` x `DirectForm2Transposed<f32>` (src/iir/biquad.rs:430-440). */
int idsp_biquad_f32_df2t_clamp(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state,
                               const float *x, float *y, size_t lanes, size_t frames,
                               int layout, void *stream);
/* `Cascade<[Biquad<f32>; n]>` x `DirectForm<f32, n>` (src/iir/biquad.rs:339-364). */
int idsp_cascade_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state,
                         const float *x, float *y, size_t lanes, size_t frames,
                         int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* iir::Biquad — f64 (same generic impls; every value is two state words,   */
/* low word first: DF1 W=8 {x0,x1,y0,y1}, DF2T W=4 {s0,s1}, Cascade 4+4n)   */
/* ------------------------------------------------------------------------ */

int idsp_biquad_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state,
                        const double *x, double *y, size_t lanes, size_t frames,
                        int layout, void *stream);
int idsp_biquad_f64_df1_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state,
                              const double *x, double *y, size_t lanes, size_t frames,
                              int layout, void *stream);
int idsp_biquad_f64_df2t(const idsp_biquad_f64 *cfg, size_t n, void *state,
                         const double *x, double *y, size_t lanes, size_t frames,
                         int layout, void *stream);
int idsp_biquad_f64_df2t_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state,
                               const double *x, double *y, size_t lanes, size_t frames,
                               int layout, void *stream);
int idsp_cascade_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state,
                         const double *x, double *y, size_t lanes, size_t frames,
                         int layout, void *stream);

/*
 * Per-lane coefficients: `ByLane<[C; N]>` (dsp-process/src/compose.rs:363-390,
 * `process_view` :375-389: lane i is filtered by configuration i with state i)
 * for C = a slice of `n` Biquad/BiquadClamp sections (compose.rs:43-77).
 *
 * `coef` is a device array of lane-contiguous planes like the state: value v
 * of section k of lane l is coef[(k * CV + v) * lanes + l], with CV = 5
 * (`Biquad::ba`, biquad.rs:116) for the plain entries and CV = 8 (ba, u, min,
 * max; biquad.rs:121-157) for the `_clamp` entries.  The i32 variants share
 * one `frac` (the const generic F of `Q32<F>`) across lanes and sections.
 * State layout, x/y layout, in-place rule and status codes are those of the
 * shared-coefficient entries above.
 */
int idsp_biquad_i32_df1_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_df1_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state,
        const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_bylane(const float *coef, size_t n, void *state,
        const float *x, float *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_clamp_bylane(const float *coef, size_t n, void *state,
        const float *x, float *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_bylane(const float *coef, size_t n, void *state,
        const float *x, float *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_clamp_bylane(const float *coef, size_t n, void *state,
        const float *x, float *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_bylane(const double *coef, size_t n, void *state,
        const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_clamp_bylane(const double *coef, size_t n, void *state,
        const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_bylane(const double *coef, size_t n, void *state,
        const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_clamp_bylane(const double *coef, size_t n, void *state,
        const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* explicit row pitches: `<entry>_pitch`                                    */
/* ------------------------------------------------------------------------ */
/*
 * Every biquad / cascade / by-lane entry above has a `_pitch` twin that takes the distance between rows of x and
 * of y explicitly (like a BLAS leading dimension), in ELEMENTS:
 *   IDSP_LANE_MAJOR   pitch = elements between the starts of consecutive lanes  (dense: frames) — the reference's
 *                     `View::<_, LaneMajor>::lane(i) = flat[i * frames ..]` (dsp-process/src/view.rs:181-195) with the
 *                     lanes padded apart; element (f, l) at l * pitch + f;
 *   IDSP_FRAME_MAJOR  pitch = elements between the starts of consecutive frames (dense: lanes): a block of `lanes`
 *                     adjacent lanes of a wider `[[T; L]; frames]` tensor; element (f, l) at f * pitch + l.
 * 0 means dense.  A pitch shorter than a row is IDSP_EINVAL; an in-place call (y == x) needs x_pitch == y_pitch.
 * Why it exists: power-of-two LANE_MAJOR pitches (frames = 4096 -> 16 KiB) alias on the HBM channels and cost
 * 10-15 % (profiles/r01_lm_pitch_probe.jsonl); a caller that owns its buffers can pad each lane by a few cache
 * lines.  The FRAME_MAJOR form lets a caller process a lane block of a larger tensor without a re-layout pass
 * (this is also how a FRAME_MAJOR host tensor is split over several devices).  `pitch == dense` is bit-identical to
 * the plain entry.
 */
int idsp_biquad_i32_df1_pitch(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_df1_clamp_pitch(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_pitch(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_clamp_pitch(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_pitch(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_clamp_pitch(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_cascade_i32_df1_pitch(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_pitch(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_clamp_pitch(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_pitch(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_clamp_pitch(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_cascade_f32_df1_pitch(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_pitch(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_clamp_pitch(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_pitch(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_clamp_pitch(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_cascade_f64_df1_pitch(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_df1_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_df1_clamp_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_dither_clamp_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_i32_wide_clamp_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch,
        int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_bylane_pitch(const float *coef, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df1_clamp_bylane_pitch(const float *coef, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_bylane_pitch(const float *coef, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f32_df2t_clamp_bylane_pitch(const float *coef, size_t n, void *state, const float *x, size_t x_pitch, float *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_bylane_pitch(const double *coef, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df1_clamp_bylane_pitch(const double *coef, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_bylane_pitch(const double *coef, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);
int idsp_biquad_f64_df2t_clamp_bylane_pitch(const double *coef, size_t n, void *state, const double *x, size_t x_pitch, double *y, size_t y_pitch,
        size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* iir::normal::Normal and iir::wdf::Wdf lanes                               */
/* ------------------------------------------------------------------------ */

/* `Normal<C>` x `DirectForm1<T>` (src/iir/normal.rs:28-58), the Rader-Gold / Chamberlin normal form:
 *   y1' = (b0*x0 + b1*x1 + b2*x2 + re*y1 + (-im)*y0).as_();  y0' = (im*y1 + re*y0).as_();  returns y0'
 * with state words {x0, x1, y0, y1} (y0 = in-phase, y1 = quadrature component, normal.rs:24-25).
 * The configuration reuses the biquad records: ba = [b0, b1, b2, p.re, p.im] (and frac for Q32<F>);
 * n sections run in series like the biquad entries. */
int idsp_normal_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y,
                        size_t lanes, size_t frames, int layout, void *stream);
int idsp_normal_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y,
                        size_t lanes, size_t frames, int layout, void *stream);
int idsp_normal_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y,
                        size_t lanes, size_t frames, int layout, void *stream);
/* `From<&[[f64; 3]; 2]> for Normal<C>` (src/iir/normal.rs:62-76): sos = [b0,b1,b2,a0,a1,a2] ->
 * out = [b0/a0, b1/a0, b2/a0, p.re, p.im]; IDSP_EINVAL when the poles are real (`assert!(pq >= 0.0)`). */
int idsp_normal_from_sos(const double sos[6], double out[5]);

/* `Wdf<N, M>` (src/iir/wdf.rs:103-171): N two-port adaptors, adaptor i of type nibble i of M
 * (`Tpa`, wdf.rs:14-32: 0xA A, 0xB B, 0xE B1, 0x1 X, 0xC C, 0xF C1, 0xD D, anything else Z), with
 * coefficients a[i] as raw `Q32<32>` bits.  State = `WdfState<N>::z` (N words per section). */
#define IDSP_WDF_MAX_ORDER 8
typedef struct idsp_wdf {
    int32_t n;   /* N: 1..8 */
    uint32_t m;  /* M: one nibble per adaptor, adaptor 0 in the low nibble */
    int32_t a[IDSP_WDF_MAX_ORDER];
} idsp_wdf;
/* `Wdf::<N, M>::quantize(&g)` (src/iir/wdf.rs:126-137, `Tpa::quantize` :50-62): fills out->a from the
 * allpass poles g[0..n) for the architecture nibbles m; IDSP_EOUTOFRANGE when a pole does not fit its
 * adaptor type (the reference returns None). */
int idsp_wdf_quantize(int n, uint32_t m, const double *g, idsp_wdf *out);
/* Per-lane state words of a chain of n_sections sections (sum of their n). */
size_t idsp_wdf_state_words(const idsp_wdf *cfg, size_t n_sections);
/* Serial chain of n_sections `Wdf` sections (array / tuple composition, dsp-process/src/compose.rs:43-113),
 * `SplitProcess<i32, i32, WdfState<N>>` (wdf.rs:153-169); in place allowed. */
int idsp_wdf_i32(const idsp_wdf *cfg, size_t n_sections, void *state, const int32_t *x, int32_t *y,
                 size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* Cic — cascaded integrator-comb rate changer (src/cic.rs)                 */
/* ------------------------------------------------------------------------ */

#define IDSP_CIC_MAX_ORDER 6
#define IDSP_CIC_MAX_DELAY 4

/* `Cic<T, N, M>::new(rate)` (src/cic.rs:13-28,39-47): order N (1..6), comb delay M
 * (1..4), rate = fast/slow - 1.  T = i32 or i64 is chosen by the entry point. */
typedef struct idsp_cic {
    int32_t order;
    int32_t comb_delay;
    uint32_t rate;
} idsp_cic;

/* `Cic::gain()` as i64 (wrapping), `gain_log2()`, `response_length()` (src/cic.rs:103-118);
 * gain_log2 / response_length return a negative idsp_status / 0 for an invalid cfg. */
int64_t idsp_cic_gain(const idsp_cic *cfg);
int idsp_cic_gain_log2(const idsp_cic *cfg);
size_t idsp_cic_response_length(const idsp_cic *cfg);
/* 32-bit state words per lane for T of `bits` (32 or 64) width; 0 for an invalid cfg.
 * Values in order: zoh, combs[N][M] (row n = comb n, oldest first), integrators[N];
 * a 64-bit value is two words, low first; planes are lane-contiguous like every state.
 * `index` is not part of the record: the chunked entries below start and end every call
 * at index == 0 (`tick()`), the only phase `Decimator` / `Interpolator` accept. */
size_t idsp_cic_state_words(const idsp_cic *cfg, int bits);

/* `Split::stateful(Cic::<T, N, M>::new(R - 1)).decimate()` (src/cic.rs:338-341; `Process<T, Option<T>>`
 * :186-207 under `Decimator`, dsp-process/src/adapters.rs:158-167): x holds `frames` chunks `[T; R]`
 * per lane (R = rate + 1; FRAME_MAJOR x[(f*lanes + l)*R + r], LANE_MAJOR x[(l*frames + f)*R + r]),
 * y one sample per chunk.  Integrators and combs wrap (cic.rs:191,199). */
int idsp_cic_dec_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);
int idsp_cic_dec_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);
/* `...interpolate()` (src/cic.rs:343-346; `Process<Option<T>, T>` :160-182 under `Interpolator`,
 * adapters.rs:27-35): one input sample per frame, y holds chunks `[T; R]`.  Release-mode
 * (wrapping) arithmetic; the reference's debug build would panic on overflow (cic.rs:177). */
int idsp_cic_int_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);
int idsp_cic_int_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* coefficient front-end (host side, no device work)                        */
/* ------------------------------------------------------------------------ */
/*
 * `iir::coefficients::Filter<T>`, `iir::pid::{Builder, Pid, Units}` and
 * `iir::config::BiquadConfig::{build, try_build}`: pure host functions that
 * produce the idsp_biquad_* records above (or rows of the per-lane coefficient
 * planes).  Parameters travel as f64; `f32 != 0` selects T = f32, i.e. every
 * field is first rounded to f32 and all arithmetic is carried out in f32 like
 * the reference's generic code instantiated with T = f32.  `validate != 0`
 * runs the reference's `validate()` first (the `try_build*` methods) and
 * returns IDSP_ENONFINITE .. IDSP_ESIGN; `validate == 0` is the unchecked
 * `build*` which, like the reference, may produce NaN/inf coefficients.
 * The `_i32` outputs are C = `Q32<frac>`, Y = i32; `_f32`: C = Y = f32;
 * `_f64`: C = Y = f64.
 */

/* `coefficients::Type` (src/iir/coefficients.rs:43-66) */
typedef enum idsp_filter_type {
    IDSP_LOWPASS = 0, IDSP_HIGHPASS, IDSP_BANDPASS, IDSP_ALLPASS, IDSP_NOTCH,
    IDSP_PEAKING, IDSP_LOWSHELF, IDSP_HIGHSHELF, IDSP_IHO
} idsp_filter_type;

/* `coefficients::Shape` (src/iir/coefficients.rs:6-16) */
typedef enum idsp_shape_kind { IDSP_SHAPE_Q = 0, IDSP_SHAPE_BANDWIDTH = 1, IDSP_SHAPE_SLOPE = 2 } idsp_shape_kind;

/* `coefficients::Filter<T>` (src/iir/coefficients.rs:27-40); Default (:88-97) is
 * frequency 0, gain 1, shelf 1, Shape::Q(1/sqrt 2). */
typedef struct idsp_filter {
    double frequency; /* angular critical frequency w0, pi = Nyquist */
    double gain;      /* linear passband gain */
    double shelf;     /* linear shelf gain (peaking / shelves / iho) */
    double shape;     /* Q, bandwidth in octaves, or shelf slope */
    int32_t shape_kind;
    int32_t f32;
} idsp_filter;

/* `Filter::build(typ)` / `try_build(typ)` (coefficients.rs:483-501): cookbook
 * `[[b0,b1,b2],[a0,a1,a2]]` flattened to ba[6] = the `sos` row format of
 * idsp_biquad_*_from_sos.  With f32 set the six values are exact f32 values. */
int idsp_filter_build(const idsp_filter *f, int type, int validate, double ba[6]);

/* `pid::Order` (src/iir/pid.rs:14-24) and `pid::Action` (:61-75) index values */
#define IDSP_PID_ORDER_P 2
#define IDSP_PID_ORDER_I 1
#define IDSP_PID_ORDER_I2 0

/* `pid::Builder<T>` (src/iir/pid.rs:40-45); gain/limit index = Action: I2, I, P, D, D2.
 * Default (:47-55): order I, gains 0, limits +inf. */
typedef struct idsp_pid_builder {
    int32_t order;
    int32_t f32;
    double gain[5];
    double limit[5];
} idsp_pid_builder;

/* `Build<[C; 5]> for Builder<T>` (src/iir/pid.rs:256-313) with context `period`;
 * validate = `Builder::validate` (:195-222).  ba = [b0, b1, b2, a1, a2]. */
int idsp_pid_build_i32(const idsp_pid_builder *b, double period, int validate, int frac, int32_t ba[5]);
int idsp_pid_build_f32(const idsp_pid_builder *b, double period, int validate, float ba[5]);
int idsp_pid_build_f64(const idsp_pid_builder *b, double period, int validate, double ba[5]);

/* `pid::Units<T>` (src/iir/pid.rs:350-369), Default 1, 1, 1 */
typedef struct idsp_units {
    double t, x, y;
} idsp_units;

/* `pid::Pid<T>` (src/iir/pid.rs:384-417); Default: builder default, setpoint 0, min -inf, max +inf */
typedef struct idsp_pid {
    idsp_pid_builder builder;
    double setpoint, min, max;
} idsp_pid;

/* `Build<BiquadClamp<C, Y>> for Pid<T>` (src/iir/pid.rs:533-567) = `BiquadConfig::Pid`
 * (config.rs:371,408); validate = `Pid::validate` (pid.rs:497-518). */
int idsp_pid_build_clamp_i32(const idsp_pid *p, const idsp_units *units, int validate, int frac,
                             idsp_biquad_clamp_i32 *out);
int idsp_pid_build_clamp_f32(const idsp_pid *p, const idsp_units *units, int validate, idsp_biquad_clamp_f32 *out);
int idsp_pid_build_clamp_f64(const idsp_pid *p, const idsp_units *units, int validate, idsp_biquad_clamp_f64 *out);

/* `config::BaConfig<T>` (src/iir/config.rs:19-31): SI-unit `[[b],[a]]`, offset, limits */
typedef struct idsp_ba_config {
    double ba[6];
    double offset, min, max;
    int32_t f32;
} idsp_ba_config;

/* `BiquadConfig::Ba(..).build(units)` (config.rs:359-367) / `.try_build` (:389-407) */
int idsp_config_ba_build_i32(const idsp_ba_config *c, const idsp_units *units, int validate, int frac,
                             idsp_biquad_clamp_i32 *out);
int idsp_config_ba_build_f32(const idsp_ba_config *c, const idsp_units *units, int validate, idsp_biquad_clamp_f32 *out);
int idsp_config_ba_build_f64(const idsp_ba_config *c, const idsp_units *units, int validate, idsp_biquad_clamp_f64 *out);

/* `config::FilterConfig<T>` (src/iir/config.rs:46-67): frequency relative to 1/units.t, gains in dB */
typedef struct idsp_filter_config {
    int32_t typ; /* idsp_filter_type */
    int32_t shape_kind;
    double frequency, gain_db, shelf_db, shape;
    double offset, min, max;
    int32_t f32;
} idsp_filter_config;

/* `BiquadConfig::Filter(..).build(units)` (config.rs:372-385) / `.try_build` (:409-427) */
int idsp_config_filter_build_i32(const idsp_filter_config *c, const idsp_units *units, int validate, int frac,
                                 idsp_biquad_clamp_i32 *out);
int idsp_config_filter_build_f32(const idsp_filter_config *c, const idsp_units *units, int validate,
                                 idsp_biquad_clamp_f32 *out);
int idsp_config_filter_build_f64(const idsp_filter_config *c, const idsp_units *units, int validate,
                                 idsp_biquad_clamp_f64 *out);

/* ------------------------------------------------------------------------ */
/* hbf — symmetric FIR and half-band decimator / interpolator cascades      */
/* ------------------------------------------------------------------------ */

#define IDSP_HBF_MAX_STAGES 5
#define IDSP_HBF_MAX_TAPS 32

/* A cascade of `EvenSymmetric<[f32; M]>` half-band stages (src/hbf.rs:70-138)
 * in PROCESSING order.  taps[s][0..m[s]) are ordered outermost (small) to
 * centre (large) exactly like HBF_TAPS (src/hbf.rs:308-349). */
typedef struct idsp_hbf_cascade_f32 {
    int32_t stages;                 /* 1..IDSP_HBF_MAX_STAGES */
    int32_t m[IDSP_HBF_MAX_STAGES]; /* one-sided tap count per stage, 1..IDSP_HBF_MAX_TAPS */
    float taps[IDSP_HBF_MAX_STAGES][IDSP_HBF_MAX_TAPS];
} idsp_hbf_cascade_f32;

/* Built-in tap sets: set 0 = HBF_TAPS (140 dB, src/hbf.rs:308-349),
 * set 1 = HBF_TAPS_98 (src/hbf.rs:258-292). */
/* Fill `out` like `HBF_DEC_CASCADE` restricted to a 2^stages rate change
 * (src/hbf.rs:385-421: highest-rate/fewest-tap stage first, i.e. tuple index
 * stages-1 down to 0). */
int idsp_hbf_dec_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out);
/* Fill `out` like `HBF_INT_CASCADE` (src/hbf.rs:476-512: tuple index 0 first). */
int idsp_hbf_int_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out);
/* `hbf_dec_response_length` / `hbf_int_response_length` (src/hbf.rs:424-448,515-539)
 * generalised to an arbitrary cascade. */
int idsp_hbf_dec_response_length(const idsp_hbf_cascade_f32 *cfg);
int idsp_hbf_int_response_length(const idsp_hbf_cascade_f32 *cfg);

/* State words per lane: for each stage s in processing order
 *   decimator   `HbfDec`  (src/hbf.rs:142-145): even[m-1] then odd[2m-1], oldest first
 *   interpolator `HbfInt` (src/hbf.rs:196-198): x[2m-1], oldest first
 * i.e. exactly the samples `copy_within` keeps (src/hbf.rs:182-183,224). */
size_t idsp_hbf_dec_state_words(const idsp_hbf_cascade_f32 *cfg);
size_t idsp_hbf_int_state_words(const idsp_hbf_cascade_f32 *cfg);

/* Half-band decimator cascade, rate change R = 2^stages
 * (`SplitProcess<[f32; R], f32, _>` for the `Major`/`ChunkIn` nest,
 * src/hbf.rs:156-192,385-421).  `frames` counts OUTPUT samples per lane; each
 * input element is a chunk `[f32; R]` of R consecutive high-rate samples:
 *   FRAME_MAJOR  x[(f*lanes + l)*R + k]   y[f*lanes + l]
 *   LANE_MAJOR   x[(l*frames + f)*R + k]  y[l*frames + f]   (lane = contiguous stream)
 */
int idsp_hbf_dec_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                     size_t lanes, size_t frames, int layout, void *stream);
/* Half-band interpolator cascade (src/hbf.rs:200-236,476-512).  `frames`
 * counts INPUT samples per lane; each output element is a chunk `[f32; R]`:
 *   FRAME_MAJOR  x[f*lanes + l]   y[(f*lanes + l)*R + k]
 *   LANE_MAJOR   x[l*frames + f]  y[(l*frames + f)*R + k]
 */
int idsp_hbf_int_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                     size_t lanes, size_t frames, int layout, void *stream);

/* Same-rate linear-phase FIR, `SplitProcess<f32, f32, [f32; N]>` for the four
 * symmetry types of `type_fir!` (src/hbf.rs:70-138, `get()` :46-68):
 *   window w of 2M + odd samples ending at the current input,
 *   y = sum_k (w[2M-1+odd-k] +/- w[k]) * tap[k]  (+ w[M] for ODD_SYMMETRIC: unity centre tap)
 * State words per lane: the last LEN = 2M - 1 + odd inputs, oldest first
 * (what `copy_within` keeps, src/hbf.rs:103,121). */
typedef enum idsp_fir_kind {
    IDSP_FIR_ODD_SYMMETRIC = 0,     /* Type I,   `OddSymmetric`      src/hbf.rs:129 */
    IDSP_FIR_EVEN_SYMMETRIC = 1,    /* Type II,  `EvenSymmetric`     src/hbf.rs:131 */
    IDSP_FIR_ODD_ANTISYMMETRIC = 2, /* Type III, `OddAntiSymmetric`  src/hbf.rs:134 */
    IDSP_FIR_EVEN_ANTISYMMETRIC = 3 /* Type IV,  `EvenAntiSymmetric` src/hbf.rs:136 */
} idsp_fir_kind;

typedef struct idsp_fir_sym_f32 {
    int32_t kind; /* idsp_fir_kind */
    int32_t m;    /* one-sided tap count, 1..IDSP_HBF_MAX_TAPS */
    float taps[IDSP_HBF_MAX_TAPS];
} idsp_fir_sym_f32;

size_t idsp_fir_sym_state_words(const idsp_fir_sym_f32 *cfg);
int idsp_fir_sym_f32_process(const idsp_fir_sym_f32 *cfg, void *state, const float *x, float *y,
                             size_t lanes, size_t frames, int layout, void *stream);

/* The same three processors on f64 samples with f64 taps: `EvenSymmetric<[C; M]>` and the `type_fir!` types are generic in
 * the sample type (src/hbf.rs:70-138, `T: Mul<C, Output = T>`), so `HbfDec<[f64; N]>` / `HbfInt<[f64; N]>` cascades run
 * with `[f64; M]` taps.  Same definitions as the f32 entries above; every f64 state value takes two state words
 * (value v -> words 2v: low half, 2v + 1: high half), so the `_f64` word counts are twice the f32 ones.  The cascade
 * builders widen the built-in f32 tap sets exactly (an f32 is an f64). */
typedef struct idsp_hbf_cascade_f64 {
    int32_t stages;
    int32_t m[IDSP_HBF_MAX_STAGES];
    double taps[IDSP_HBF_MAX_STAGES][IDSP_HBF_MAX_TAPS];
} idsp_hbf_cascade_f64;
typedef struct idsp_fir_sym_f64 {
    int32_t kind; /* idsp_fir_kind */
    int32_t m;
    double taps[IDSP_HBF_MAX_TAPS];
} idsp_fir_sym_f64;
int idsp_hbf_dec_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out);
int idsp_hbf_int_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out);
size_t idsp_hbf_dec_state_words_f64(const idsp_hbf_cascade_f64 *cfg);
size_t idsp_hbf_int_state_words_f64(const idsp_hbf_cascade_f64 *cfg);
size_t idsp_fir_sym_state_words_f64(const idsp_fir_sym_f64 *cfg);
int idsp_hbf_dec_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y,
                     size_t lanes, size_t frames, int layout, void *stream);
int idsp_hbf_int_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y,
                     size_t lanes, size_t frames, int layout, void *stream);
int idsp_fir_sym_f64_process(const idsp_fir_sym_f64 *cfg, void *state, const double *x, double *y,
                             size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* cossin / Accu DDS / Lockin                                               */
/* ------------------------------------------------------------------------ */

/* `cossin(phase)` (src/cossin.rs:14-67) elementwise: out[2*i] = cos, out[2*i+1]
 * = sin — the `cossin(p) -> i32[N,2]` function of src/py.rs:10-28. */
int idsp_cossin_i32(const int32_t *phase, int32_t *out, size_t n, void *stream);

/* `atan2(y, x)` (src/atan2.rs:66-82) elementwise over rows [x, y] — the
 * `atan2(xy: i32[N,2]) -> i32[N]` function of src/py.rs:30-47 and `Complex::<i32>::arg`
 * (src/complex.rs:254-256) for `Complex<i32>` = [re, im] rows such as the lock-in output.
 * Result: i32::MIN = -pi ... i32::MAX = one count below +pi. */
int idsp_atan2_i32(const int32_t *xy, int32_t *out, size_t n, void *stream);

/* DDS: per lane `Accu<Wrapping<i32>>` (src/accu.rs:34-41, pre-increment) feeding
 * `Complex::<i32>::from_angle` (src/complex.rs:237-240).  State words per lane:
 * { accu.state, accu.step }.  Output element = Complex<i32> = [re, im] adjacent
 * (src/complex.rs:15-20): out[(index(f,l))*2 + {0,1}]. */
int idsp_dds_i32(void *state, int32_t *out, size_t lanes, size_t frames, int layout,
                 void *stream);

#define IDSP_LOCKIN_MAX_CASCADE 4

/* `Lockin<[Lowpass<N>; K]>` (src/lockin.rs:11-39, src/lowpass.rs:13-80).
 * order = N (1 or 2), cascade = K (1..IDSP_LOCKIN_MAX_CASCADE),
 * k[c][0..N) = `Lowpass<N>.0` of cascade element c. */
typedef struct idsp_lockin_i32 {
    int32_t order;
    int32_t cascade;
    int32_t k[IDSP_LOCKIN_MAX_CASCADE][2];
} idsp_lockin_i32;

/* State words per lane: { accu.state, accu.step } followed by
 * `[[LowpassState<N>; K]; 2]` (index 0 = I/re, 1 = Q/im), each i64 as lo,hi:
 *   word 2 + ((iq*K + c)*N + j)*2 + {0: lo, 1: hi}. */
size_t idsp_lockin_state_words(const idsp_lockin_i32 *cfg);

/* Phase-accumulator lock-in: per frame `phase = accu.next()`, then
 * `Lockin::process(state, (x, Wrapping(phase)))` (src/lockin.rs:30-39):
 * cossin -> `x * Q32<32>` mix (dsp-fixedpoint/src/lib.rs:449-456) -> the lowpass
 * cascade on I and Q.  Output Complex<i32>: y[index(f,l)*2 + {0: re, 1: im}]. */
int idsp_lockin_i32_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x,
                            int32_t *y, size_t lanes, size_t frames, int layout,
                            void *stream);

/* The lock-in above with the polar read-out of its `Complex<i32>` output fused in (SURVEY 8f rank 2:
 * cossin -> mix -> lowpass -> atan2 in one pass, no Complex<i32> round trip through memory):
 *   _arg:      y[index(f,l)] = `Lockin::process(..).arg()`       (src/complex.rs:254-256 -> src/atan2.rs:66-82), i32
 *   _norm_sqr: y[index(f,l)] = `Lockin::process(..).norm_sqr()`  (src/complex.rs:214-217), i64; the sum wraps
 *              for (i32::MIN, i32::MIN) as in a release build.
 * Same cfg and state as idsp_lockin_i32_process; the three entries may be mixed on one state. */
int idsp_lockin_i32_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                        size_t lanes, size_t frames, int layout, void *stream);
int idsp_lockin_i32_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x,
                             int64_t *y, size_t lanes, size_t frames, int layout, void *stream);

/* ---- `Lockin<C>` as the reference defines it (src/lockin.rs:11-39): any arm filter `C: SplitProcess<X, X, S>`, the LO
 * either from a phase (:30-39, the entries above and idsp_lockin_i32_biquad_process) or given per sample (:17-27, the `_lo`
 * entries).  Arm filters here: `[Lowpass<N>; K]` (idsp_lockin_i32 above) and `[Biquad<C>; n]` x `[DirectForm1<T>; n]`
 * (src/iir/biquad.rs:366-383 under the array composition of dsp-process/src/compose.rs:80-113; n = 1 is a plain
 * `Biquad<C>`), 1 <= n <= IDSP_LOCKIN_MAX_SECTIONS; the same sections run on I (`state[0]`) and on Q (`state[1]`).
 * External LO: lo[index(f,l)*2 + {0: re, 1: im}] holds `Complex<Q32<32>>` bits (i32) or `Complex<f32>`; the mixer is
 * `x * lo.re`, `x * lo.im` — for i32 `((q as i64 * x as i64) >> 32) as i32` (dsp-fixedpoint/src/lib.rs:449-456), for f32 one
 * rounded multiply — and the output `Complex<X>` = [re, im] adjacent as above.  x, lo and y are three separate buffers
 * (overlap is IDSP_EINVAL), lo and y 8-byte aligned (they hold pairs).
 * The f32 entry with lo = (cos, -sin) is the `mix * lowpass.lanes()` graph of examples/ddc_lockin.rs:35-42.
 * State words per lane: phase form { accu.state, accu.step, I: n x {x0,x1,y0,y1}, Q: n x {x0,x1,y0,y1} };
 * `_lo` forms: the two arms only — biquad arms { I: n x {x0,x1,y0,y1}, Q: ... }, lowpass arms the words of
 * idsp_lockin_state_words() without its first two (idsp_lockin_state_words(cfg) - 2). */
#define IDSP_LOCKIN_MAX_SECTIONS 4
size_t idsp_lockin_biquad_state_words(size_t n, int with_accu);
int idsp_lockin_i32_biquad_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x, int32_t *y,
                                   size_t lanes, size_t frames, int layout, void *stream);
int idsp_lockin_i32_lo_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y,
                               size_t lanes, size_t frames, int layout, void *stream);
int idsp_lockin_i32_biquad_lo_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x,
                                      const int32_t *lo, int32_t *y, size_t lanes, size_t frames, int layout, void *stream);
int idsp_lockin_f32_biquad_lo_process(const idsp_biquad_f32 *sections, size_t n, void *state, const float *x,
                                      const float *lo, float *y, size_t lanes, size_t frames, int layout, void *stream);

/* `Lowpass<N>` cascade alone on a real stream (src/lowpass.rs:47-78; array
 * composition dsp-process/src/compose.rs:80-113).  State: `[LowpassState<N>; K]`
 * words ((c*N + j)*2 + {lo,hi}). */
int idsp_lowpass_i32(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);

/* FM discriminator receiver core of examples/fm_disc.rs:25-50, `(disc * deemph).minor()`:
 *   z = x * conj(prev.into_bits())   (`Complex<Q32<32>> * Complex<i32>`, src/complex.rs:117-134:
 *        re = (x.re*p.re - x.im*(-p.im)) >> 32, im = (x.re*(-p.im) + x.im*p.re) >> 32, i64 wrapping)
 *   d = z.arg() - carrier            (src/complex.rs:254-256 -> atan2, wrapping; 0 while prev is None)
 *   y = deemph(d)                    (`Biquad<Q32<F>>` x `DirectForm1`, src/iir/biquad.rs:366-383)
 * x holds `Complex<Q32<32>>` bits [re, im] per sample (8 bytes), y one i32 per sample.
 * State words per lane: {has_prev, prev.re, prev.im, x0, x1, y0, y1}; zero = (None, DirectForm1::default()). */
typedef struct idsp_fm_disc {
    int32_t carrier;
    idsp_biquad_i32 deemph;
} idsp_fm_disc;
#define IDSP_FM_DISC_STATE_WORDS 7
int idsp_fm_disc_i32(const idsp_fm_disc *cfg, void *state, const int32_t *x, int32_t *y,
                     size_t lanes, size_t frames, int layout, void *stream);

/* ------------------------------------------------------------------------ */
/* lane split over several devices in ONE process: idsp_multi_*             */
/* ------------------------------------------------------------------------ */
/*
 * Lanes never interact (`Lanes::process` indexes state[i], x[i] only, dsp-process/src/compose.rs:468-476), so G
 * devices take G contiguous lane blocks — device g gets lanes [g*L/G, (g+1)*L/G) — and there is NO data-path
 * exchange between them.  An idsp_multi holds a device list and one non-blocking stream per device; it is host-side
 * bookkeeping for hosts that drive all GPUs from one process (the Rust shim, a C program).  One process per GPU
 * (torch.distributed / MPI, each calling the plain entry points on its own lane block) is equivalent and is what
 * bench.py uses.  Per-device buffers hold that device's lane block only: LANE_MAJOR shards are contiguous pieces of
 * the host tensor, FRAME_MAJOR shards are `[[T; L/G]; frames]` tensors of their own (or lane blocks of a wider
 * tensor through the `_pitch` entries).
 */
typedef struct idsp_multi idsp_multi;

/* `devices` = n_devices device ordinals (a device may appear more than once: its lane blocks then share the
 * device); devices == NULL and n_devices <= 0 = every visible device. */
int idsp_multi_create(const int *devices, int n_devices, idsp_multi **out);
int idsp_multi_destroy(idsp_multi *m);
/* Number of lane blocks G; device ordinal and stream (hipStream_t) of block `index`. */
int idsp_multi_size(const idsp_multi *m);
int idsp_multi_device(const idsp_multi *m, int index);
void *idsp_multi_stream(const idsp_multi *m, int index);
/* Lane block [*lane_lo, *lane_hi) of block `index` for a job of `lanes` lanes. */
int idsp_multi_shard(const idsp_multi *m, size_t lanes, int index, size_t *lane_lo, size_t *lane_hi);

/* Generic driver: for every block, make its device current and call fn(user, index, lane_lo, lane_hi, stream) —
 * fn issues whatever entry points it likes on that stream for that lane block.  ANY non-zero return stops the
 * loop and is returned as it is (negative: an idsp_status; positive: the callback's own stop code).  Blocks before
 * the stopping one have been handed to fn already and their work is in flight: after a stop the per-block state is
 * indeterminate until the caller has synchronised and reloaded it; idsp_multi_last_block() names the block.
 * The caller's current device is restored. */
typedef int (*idsp_shard_fn)(void *user, int index, size_t lane_lo, size_t lane_hi, void *stream);
int idsp_multi_for_each(idsp_multi *m, size_t lanes, idsp_shard_fn fn, void *user);
/* Index of the lane block at which the most recent idsp_multi_for_each / idsp_multi_biquad_* call of this thread
 * stopped with a non-zero status; -1 if it ran through. */
int idsp_multi_last_block(void);
/* Wait for every block's stream (the "barrier" of a single-process run). */
int idsp_multi_sync(idsp_multi *m);

/* Per-block device buffers of (block lanes) * bytes_per_lane bytes, zero-filled: ptrs[G] out. */
int idsp_multi_alloc(idsp_multi *m, size_t lanes, size_t bytes_per_lane, void **ptrs);
int idsp_multi_free(idsp_multi *m, void **ptrs);
/* Scatter (to_device != 0) / gather a host array of `lanes` records of bytes_per_lane bytes, lane-contiguous (a
 * LANE_MAJOR tensor, or one state / coefficient plane), to / from the per-block buffers; asynchronous on the
 * block streams (pinned host memory overlaps). */
int idsp_multi_copy(idsp_multi *m, size_t lanes, size_t bytes_per_lane, void *const *dev_ptrs, void *host, int to_device);

/* The two headline operators over the split: state[g], x[g], y[g] are block g's device buffers; `lanes` is the
 * TOTAL lane count.  Every block's arguments are validated BEFORE any block is launched (a bad block is reported
 * with nothing in flight); a failure after that leaves earlier blocks running and the state indeterminate, see
 * idsp_multi_last_block().  Asynchronous; idsp_multi_sync() waits. */
int idsp_multi_biquad_i32_df1(idsp_multi *m, const idsp_biquad_i32 *cfg, size_t n, void *const *state,
                              const int32_t *const *x, int32_t *const *y, size_t lanes, size_t frames, int layout);
int idsp_multi_biquad_f32_df2t(idsp_multi *m, const idsp_biquad_f32 *cfg, size_t n, void *const *state,
                               const float *const *x, float *const *y, size_t lanes, size_t frames, int layout);

#ifdef __cplusplus
}
#endif
#endif /* IDSP_HIP_H */
