/*
 * idsp_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference algorithms on the hot path
 * (quartiq/idsp 0.22.0).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; nothing under
 * idsp_amd/ may include, link or call it.
 *
 * Every `idsp_ref_*` function is the twin of the `idsp_*` entry point of the
 * same name in include/idsp_hip.h: identical arguments (minus `stream`),
 * identical buffer and state layouts, but HOST pointers, executed by a scalar
 * loop that follows the reference line by line (citations at each function in
 * idsp_oracle.c).
 *
 * Pinning status:
 *   pinned by reference known-answer tests (tests/golden/ref_kat.json):
 *     i32 DF1 biquad + float->Q quantisation, BiquadClamp, DF2T identity,
 *     DF1Dither doctest, HbfDec KAT + response lengths, cossin error bounds,
 *     Accu doctest, Lanes/LaneMajor view semantics; atan2 (exact zero-axis
 *     values + error bounds over the reference's grid); Cic (the reference's
 *     quickcheck properties and its "Cic == Integrator^N -> Downsample ->
 *     Comb^N" tests); the fm_disc example's own test (corr / gain / rms).
 *   PARITY UNPINNED (no asserted value exists in the reference): Lowpass<1|2>,
 *     Lockin (and its fused arg / norm_sqr read-outs = Lockin followed by the
 *     KAT-pinned atan2 / a wrapping sum of squares), DirectForm1Wide, clamp on
 *     Dither/Wide, HbfInt sample values,
 *     HBF_TAPS_98, exact cossin outputs at given phases, ByLane (pinned only
 *     through "lane i == the pinned shared-coefficient entry run alone"),
 *     Normal and Wdf (no test module in the reference).  For these the pin is
 *     the agreement of two independent restatements (this file and
 *     oracle/spec.py) plus properties that follow from the cited lines.
 * The reference itself is Rust and cannot be built in this image (no
 * rustc/cargo), so there is no oracle/_ref.
 */
#ifndef IDSP_ORACLE_H
#define IDSP_ORACLE_H

#include "../include/idsp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* scalar helpers exported for direct known-answer tests */
void idsp_ref_cossin(int32_t phase, int32_t *cos_out, int32_t *sin_out);
const uint32_t *idsp_ref_cossin_table(void); /* 128 entries, build.rs:8-41 */
int32_t idsp_ref_quantize_f64(double v, int frac);
int idsp_ref_biquad_i32_from_sos(const double sos[6], int frac, idsp_biquad_i32 *out);
int idsp_ref_biquad_f32_from_sos(const float sos[6], idsp_biquad_f32 *out);
int idsp_ref_biquad_f32_from_sos_f64(const double sos[6], idsp_biquad_f32 *out);
/* coefficients::Filter::{lowpass,highpass} in f64 (src/iir/coefficients.rs:266-335);
 * w0 = angular critical frequency, q = Shape::Q. Output sos = [b0,b1,b2,a0,a1,a2]. */
void idsp_ref_filter_lowpass(double w0, double gain, double q, double sos[6]);
void idsp_ref_filter_highpass(double w0, double gain, double q, double sos[6]);

int idsp_ref_biquad_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state,
                            const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_df1_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                                  const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_dither(const idsp_biquad_i32 *cfg, size_t n, void *state,
                               const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_dither_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                                     const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_wide(const idsp_biquad_i32 *cfg, size_t n, void *state,
                             const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_wide_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state,
                                   const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cascade_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state,
                             const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);

int idsp_ref_biquad_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state,
                            const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df1_clamp(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state,
                                  const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df2t(const idsp_biquad_f32 *cfg, size_t n, void *state,
                             const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df2t_clamp(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state,
                                   const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cascade_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state,
                             const float *x, float *y, size_t lanes, size_t frames, int layout);

int idsp_ref_biquad_f64_from_sos(const double sos[6], idsp_biquad_f64 *out);
int idsp_ref_biquad_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state,
                            const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df1_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state,
                                  const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df2t(const idsp_biquad_f64 *cfg, size_t n, void *state,
                             const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df2t_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state,
                                   const double *x, double *y, size_t lanes, size_t frames, int layout);
/* ByLane<[C; N]> (dsp-process/src/compose.rs:363-390) */
int idsp_ref_biquad_i32_df1_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_df1_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_dither_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_dither_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_wide_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_i32_wide_clamp_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df1_bylane(const float *coef, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df1_clamp_bylane(const float *coef, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df2t_bylane(const float *coef, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f32_df2t_clamp_bylane(const float *coef, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df1_bylane(const double *coef, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df1_clamp_bylane(const double *coef, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df2t_bylane(const double *coef, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_biquad_f64_df2t_clamp_bylane(const double *coef, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cascade_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state,
                             const double *x, double *y, size_t lanes, size_t frames, int layout);

int idsp_ref_hbf_dec_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out);
int idsp_ref_hbf_int_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out);
int idsp_ref_hbf_dec_response_length(const idsp_hbf_cascade_f32 *cfg);
int idsp_ref_hbf_int_response_length(const idsp_hbf_cascade_f32 *cfg);
size_t idsp_ref_hbf_dec_state_words(const idsp_hbf_cascade_f32 *cfg);
size_t idsp_ref_hbf_int_state_words(const idsp_hbf_cascade_f32 *cfg);
int idsp_ref_hbf_dec_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout);
int idsp_ref_hbf_int_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout);

size_t idsp_ref_fir_sym_state_words(const idsp_fir_sym_f32 *cfg);
int idsp_ref_fir_sym_f32_process(const idsp_fir_sym_f32 *cfg, void *state, const float *x, float *y,
                                 size_t lanes, size_t frames, int layout);

/* Normal (src/iir/normal.rs) and Wdf (src/iir/wdf.rs) */
int idsp_ref_normal_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_normal_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_normal_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_normal_from_sos(const double sos[6], double out[5]);
int idsp_ref_wdf_quantize(int n, uint32_t m, const double *g, idsp_wdf *out);
size_t idsp_ref_wdf_state_words(const idsp_wdf *cfg, size_t n_sections);
int idsp_ref_wdf_i32(const idsp_wdf *cfg, size_t n_sections, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
/* Cic (src/cic.rs) */
int64_t idsp_ref_cic_gain(const idsp_cic *cfg);
int idsp_ref_cic_gain_log2(const idsp_cic *cfg);
size_t idsp_ref_cic_response_length(const idsp_cic *cfg);
size_t idsp_ref_cic_state_words(const idsp_cic *cfg, int bits);
int idsp_ref_cic_dec_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cic_dec_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cic_int_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_cic_int_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_fm_disc_i32(const idsp_fm_disc *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout);
int32_t idsp_ref_atan2(int32_t y, int32_t x);
int idsp_ref_atan2_i32(const int32_t *xy, int32_t *out, size_t n);
int idsp_ref_cossin_i32(const int32_t *phase, int32_t *out, size_t n);
int idsp_ref_dds_i32(void *state, int32_t *out, size_t lanes, size_t frames, int layout);
int idsp_ref_hbf_dec_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out);
int idsp_ref_hbf_int_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out);
size_t idsp_ref_hbf_dec_state_words_f64(const idsp_hbf_cascade_f64 *cfg);
size_t idsp_ref_hbf_int_state_words_f64(const idsp_hbf_cascade_f64 *cfg);
size_t idsp_ref_fir_sym_state_words_f64(const idsp_fir_sym_f64 *cfg);
int idsp_ref_hbf_dec_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_hbf_int_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
int idsp_ref_fir_sym_f64_process(const idsp_fir_sym_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout);
size_t idsp_ref_lockin_state_words(const idsp_lockin_i32 *cfg);
int idsp_ref_lockin_i32_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x,
                                int32_t *y, size_t lanes, size_t frames, int layout);
size_t idsp_ref_lockin_biquad_state_words(size_t n, int with_accu);
int idsp_ref_lockin_i32_biquad_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x, int32_t *y,
                                       size_t lanes, size_t frames, int layout);
int idsp_ref_lockin_i32_lo_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y,
                                   size_t lanes, size_t frames, int layout);
int idsp_ref_lockin_i32_biquad_lo_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x,
                                          const int32_t *lo, int32_t *y, size_t lanes, size_t frames, int layout);
int idsp_ref_lockin_f32_biquad_lo_process(const idsp_biquad_f32 *sections, size_t n, void *state, const float *x,
                                          const float *lo, float *y, size_t lanes, size_t frames, int layout);
int idsp_ref_lockin_i32_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                            size_t lanes, size_t frames, int layout);
int idsp_ref_lockin_i32_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int64_t *y,
                                 size_t lanes, size_t frames, int layout);
int idsp_ref_lowpass_i32(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                         size_t lanes, size_t frames, int layout);

/* CPU-baseline helpers for bench.py (and the full-size GPU tests): the lanes split into `threads`
 * contiguous blocks, one pinned POSIX thread each.  LANE_MAJOR blocks run the reference's serial lane loop
 * (dsp-process/src/compose.rs:478-494), FRAME_MAJOR blocks its frames-outer `[X; N]` loop
 * (compose.rs:468-476 under process.rs:122-127).  `_mt_reps`: kind 0 = biquad_i32_df1, 1 = biquad_f32_df2t;
 * each thread repeats its block `reps` times (state carried) so thread start-up is paid once. */
int idsp_ref_host_cpus(void);
int idsp_ref_biquad_mt_reps(int kind, const void *cfg, size_t n, void *state, const void *x, void *y,
                            size_t lanes, size_t frames, int layout, int threads, int reps);
int idsp_ref_biquad_i32_df1_mt(const idsp_biquad_i32 *cfg, size_t n, void *state,
                               const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                               int layout, int threads);
int idsp_ref_biquad_f32_df2t_mt(const idsp_biquad_f32 *cfg, size_t n, void *state,
                                const float *x, float *y, size_t lanes, size_t frames,
                                int layout, int threads);

#ifdef __cplusplus
}
#endif
#endif
