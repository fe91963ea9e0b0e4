// hbf.hip — half-band FIR decimator / interpolator cascades over many lanes
// (reference: src/hbf.rs — `get()` :46-68, `HbfDec` :142-192, `HbfInt`
// :196-236, cascades :385-421,476-512).
//
// Mapping.  One 256-thread workgroup owns one lane (one contiguous stream in
// LANE_MAJOR) and walks it in chunks of kChunk high-rate samples.  All stages
// of the cascade run back to back inside the chunk with the inter-stage
// streams held in LDS — the role of the reference's `Major` scratch buffers
// (dsp-process/src/compose.rs:581-593) — so HBM sees each input sample once
// and each output sample once.  The per-stage history that the reference
// keeps at the front of its state arrays (`copy_within`, src/hbf.rs:182-183,
// :224) sits in front of each LDS stream buffer and is rolled after every
// chunk; it is loaded from / stored to the caller's state words at entry /
// exit, so chunked calls continue bit-exactly.
//
// Arithmetic is the reference's, operation for operation: for each output the
// window sum  Σ_k (new_k + old_k) * tap_k  is accumulated sequentially from
// the outermost tap (k = 0) inwards starting from -0.0 (`f32::sum`), then the
// delayed even sample is added.  Compiled with -ffp-contract=off.
//
// LDS layout of one stream buffer with history length H:
//   [ pad d = (4 - H % 4) % 4 | H history words | new samples ... | slack ]
// so the new samples start 16-byte aligned (wide aligned writes) and a thread
// that owns outputs i0..i0+3 (i0 % 4 == 0) reads its window with aligned
// ds_read_b128 from word i0 and addresses it with the compile-time offset d.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "hbf_taps.h"

namespace idsp {

bool hbf_cfg_ok(const idsp_hbf_cascade_f32 *c);  // api_util.hip
// hbf_wave_{dec,int}.hip: statically specialised kernels for the built-in cascades (0 = launched)
int hbf_wave_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);
int hbf_wave_int(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);
// hbf_ring_dec.hip: LDS-DMA ring decimators (0 = launched, 1 = shape not covered, 2 = HIP error)
// hbf_blk_dec.hip: register-blocked decimators (same return codes)
int hbf_blk_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                bool lane_major, hipStream_t stream);
int hbf_ring_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4096;  // high-rate samples per chunk and lane
constexpr int kSlack = 16;    // words readable past the last valid sample

struct HbfArgs {
    int32_t stages;
    int32_t m[IDSP_HBF_MAX_STAGES];
    int32_t buf_a[IDSP_HBF_MAX_STAGES];  // LDS word offset: dec even stream / int x stream
    int32_t buf_b[IDSP_HBF_MAX_STAGES];  // LDS word offset: dec odd stream
    int32_t st_off[IDSP_HBF_MAX_STAGES + 1];  // state word offset of each stage (+ total)
    int32_t stage_off;                        // LDS word offset: FRAME_MAJOR output staging
    int32_t ablate;                           // DEBUG: bit s set = skip stage s arithmetic (timing ablation only)
    float taps[IDSP_HBF_MAX_STAGES][IDSP_HBF_MAX_TAPS];
};

// native clang vectors: one ds_read_b128 / ds_write_b128 / global dwordx4 per access (HIP's
// float4 is a struct whose member-wise copies the compiler splits into ds_read2_b32 pairs,
// which bank-conflict 4-way at a 16-byte lane stride)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int pad4(int h) { return (4 - h % 4) % 4; }
__host__ __device__ constexpr int up4(int v) { return (v + 3) & ~3; }

// Σ_k (w[hi - k] + w[lo + k]) * c[k], k = 0..M-1, sequential from -0.0  (src/hbf.rs:60-66)
template <int M>
__device__ __forceinline__ float window_sum(const float *w, int lo, const float (&c)[IDSP_HBF_MAX_TAPS])
{
    float acc = -0.0f;
#pragma unroll
    for (int k = 0; k < M; k++) acc = acc + (w[lo + 2 * M - 1 - k] + w[lo + k]) * c[k];
    return acc;
}

// ---------------------------------------------------------------- decimator
// One `HbfDec` stage (src/hbf.rs:163-185) for n outputs.
//   E: even stream, history He = M-1;  O: odd stream, history Ho = 2M-1
//   y[i] = get(O)[i] + E[i]   (logical indices, 0 = oldest history word)
// Outputs go either to the next stage's streams as pairs [even, odd] or to
// global memory.
template <int M>
__device__ __forceinline__ void dec_stage(const float *E, const float *O, int n, const float (&taps)[IDSP_HBF_MAX_TAPS],
                                          float *En, float *On, float *yg, int tid)
{
    constexpr int de = pad4(M - 1), dq = pad4(2 * M - 1);
    constexpr int WO = up4(dq + 2 * M + 3), WE = up4(de + 4);
    const int nq = (n + 3) >> 2;
    for (int g = tid; g < nq; g += kThreads) {
        const int i0 = g * 4;
        float w[WO], e[WE];
#pragma unroll
        for (int v = 0; v < WO / 4; v++) {
            const v4f t = *reinterpret_cast<const v4f *>(O + i0 + 4 * v);
            w[4 * v] = t.x, w[4 * v + 1] = t.y, w[4 * v + 2] = t.z, w[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int v = 0; v < WE / 4; v++) {
            const v4f t = *reinterpret_cast<const v4f *>(E + i0 + 4 * v);
            e[4 * v] = t.x, e[4 * v + 1] = t.y, e[4 * v + 2] = t.z, e[4 * v + 3] = t.w;
        }
        float out[4];
#pragma unroll
        for (int p = 0; p < 4; p++) out[p] = window_sum<M>(w, dq + p, taps) + e[de + p];
        if (yg) {
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (i0 + p < n) yg[i0 + p] = out[p];
        } else {
            // `ChunkIn<_, 2>`: consecutive outputs pair up as the next [even, odd]
            *reinterpret_cast<v2f *>(En + (i0 >> 1)) = v2f{out[0], out[2]};
            *reinterpret_cast<v2f *>(On + (i0 >> 1)) = v2f{out[1], out[3]};
        }
    }
}

// runtime-M fallback (arbitrary tap counts): one output per thread and pass
__device__ void dec_stage_any(int M, const float *E, const float *O, int n, const float *taps, float *En, float *On,
                              float *yg, int tid)
{
    const int de = pad4(M - 1), dq = pad4(2 * M - 1);
    for (int i = tid; i < n; i += kThreads) {
        float acc = -0.0f;
        for (int k = 0; k < M; k++) acc = acc + (O[dq + i + 2 * M - 1 - k] + O[dq + i + k]) * taps[k];
        const float out = acc + E[de + i];
        if (yg)
            yg[i] = out;
        else if (i & 1)
            On[i >> 1] = out;
        else
            En[i >> 1] = out;
    }
}

__device__ __forceinline__ void dec_stage_dispatch(int M, const float *E, const float *O, int n,
                                                   const float (&taps)[IDSP_HBF_MAX_TAPS], float *En, float *On,
                                                   float *yg, int tid)
{
    switch (M) {
        case 2: dec_stage<2>(E, O, n, taps, En, On, yg, tid); break;
        case 3: dec_stage<3>(E, O, n, taps, En, On, yg, tid); break;
        case 4: dec_stage<4>(E, O, n, taps, En, On, yg, tid); break;
        case 5: dec_stage<5>(E, O, n, taps, En, On, yg, tid); break;
        case 6: dec_stage<6>(E, O, n, taps, En, On, yg, tid); break;
        case 10: dec_stage<10>(E, O, n, taps, En, On, yg, tid); break;
        case 15: dec_stage<15>(E, O, n, taps, En, On, yg, tid); break;
        case 23: dec_stage<23>(E, O, n, taps, En, On, yg, tid); break;
        default: dec_stage_any(M, E, O, n, taps, En, On, yg, tid); break;
    }
}

__global__ __launch_bounds__(kThreads) void hbf_dec_kernel(const HbfArgs a, uint32_t *st, const float *x, float *y,
                                                           const size_t lanes, const size_t frames, const int lane_major)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    // FRAME_MAJOR: neighbouring lanes share cache lines -> give each XCD (own L2) a contiguous eighth of the lanes
    const size_t lane = lane_major ? size_t(blockIdx.x) : (blockIdx.x % 8) * ((lanes + 7) / 8) + blockIdx.x / 8;
    if (lane >= lanes) return;
    const int S = a.stages;
    const int R = 1 << S;

    // history <- state words (per stage: even[M-1] then odd[2M-1], oldest first)
    for (int s = 0; s < S; s++) {
        const int M = a.m[s], He = M - 1, Ho = 2 * M - 1;
        float *E = lds + a.buf_a[s] + pad4(He), *O = lds + a.buf_b[s] + pad4(Ho);
        for (int w = tid; w < He + Ho; w += kThreads) {
            const float v = __uint_as_float(st[size_t(a.st_off[s] + w) * lanes + lane]);
            if (w < He)
                E[w] = v;
            else
                O[w - He] = v;
        }
    }

    const size_t ch = size_t(kChunk >> S);  // output frames per chunk
    const int M0 = a.m[0];
    float *E0n = lds + a.buf_a[0] + up4(M0 - 1), *O0n = lds + a.buf_b[0] + up4(2 * M0 - 1);
    const bool vec4 = lane_major && ((frames * size_t(R)) % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    float *ystage = lds + a.stage_off;

    // Register prefetch of the next chunk (vec4 path): the global loads of chunk c+1 are
    // issued right after chunk c has been handed to LDS, so their HBM latency overlaps the
    // stage arithmetic instead of stalling the workgroup at the top of every chunk.
    constexpr int kPre = kChunk / 4 / kThreads;  // float4 pieces per thread and chunk
    v4f pre[kPre];
    auto fetch = [&](size_t f0) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch);
        const v4f *x4 = reinterpret_cast<const v4f *>(x + (lane * frames + f0) * size_t(R));
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int q = tid + i * kThreads;
            if (q < nf * R / 4) pre[i] = x4[q];
        }
    };
    if (vec4) fetch(0);

    for (size_t f0 = 0; f0 < frames; f0 += ch) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch);
        const int nin = nf * R;
        // stage-0 input: pairs [even, odd] split into the two streams
        if (vec4) {
#pragma unroll
            for (int i = 0; i < kPre; i++) {
                const int q = tid + i * kThreads;
                if (q < nin / 4) {
                    *reinterpret_cast<v2f *>(E0n + 2 * q) = v2f{pre[i].x, pre[i].z};
                    *reinterpret_cast<v2f *>(O0n + 2 * q) = v2f{pre[i].y, pre[i].w};
                }
            }
            if (f0 + ch < frames) fetch(f0 + ch);
        } else {
            const int ppf = R / 2;  // pairs per frame
            for (int q = tid; q < nin / 2; q += kThreads) {
                const size_t f = f0 + size_t(q / ppf);
                const size_t base = lane_major ? (lane * frames + f) * size_t(R) : (f * lanes + lane) * size_t(R);
                const v2f v = *reinterpret_cast<const v2f *>(x + base + size_t(q % ppf) * 2);
                E0n[q] = v.x;
                O0n[q] = v.y;
            }
        }
        lds_barrier();

        int n = nin;
        for (int s = 0; s < S; s++) {
            n >>= 1;
            const int M = a.m[s];
            const float *E = lds + a.buf_a[s], *O = lds + a.buf_b[s];
#ifdef IDSP_DEBUG_ABLATE
            if (a.ablate & (1 << s)) {
                lds_barrier();
                continue;
            }
#endif
            if (s + 1 < S) {
                const int Mn = a.m[s + 1];
                dec_stage_dispatch(M, E, O, n, a.taps[s], lds + a.buf_a[s + 1] + up4(Mn - 1),
                                   lds + a.buf_b[s + 1] + up4(2 * Mn - 1), nullptr, tid);
            } else {
                float *yg = lane_major ? y + lane * frames + f0 : nullptr;
                if (lane_major) {
                    dec_stage_dispatch(M, E, O, n, a.taps[s], nullptr, nullptr, yg, tid);
                } else {
                    // FRAME_MAJOR outputs are `lanes` apart: stage them in LDS first
                    dec_stage_dispatch(M, E, O, n, a.taps[s], nullptr, nullptr, ystage, tid);
                }
            }
            lds_barrier();
        }
        if (!lane_major) {
            for (int i = tid; i < nf; i += kThreads) y[(f0 + size_t(i)) * lanes + lane] = ystage[i];
            lds_barrier();
        }

        // roll the histories: word j <- word n_s + j (src/hbf.rs:182-183)
        float keep[2];
        int c = 0;
        for (int w = tid; w < a.st_off[S]; w += kThreads, c++) {
            int s = 0;
            while (w >= a.st_off[s + 1]) s++;
            const int M = a.m[s], He = M - 1, j = w - a.st_off[s];
            const int ns = nin >> (s + 1);
            keep[c & 1] = j < He ? lds[a.buf_a[s] + pad4(He) + ns + j] : lds[a.buf_b[s] + pad4(2 * M - 1) + ns + (j - He)];
        }
        lds_barrier();
        c = 0;
        for (int w = tid; w < a.st_off[S]; w += kThreads, c++) {
            int s = 0;
            while (w >= a.st_off[s + 1]) s++;
            const int M = a.m[s], He = M - 1, j = w - a.st_off[s];
            if (j < He)
                lds[a.buf_a[s] + pad4(He) + j] = keep[c & 1];
            else
                lds[a.buf_b[s] + pad4(2 * M - 1) + (j - He)] = keep[c & 1];
        }
        lds_barrier();
    }

    for (int w = tid; w < a.st_off[S]; w += kThreads) {
        int s = 0;
        while (w >= a.st_off[s + 1]) s++;
        const int M = a.m[s], He = M - 1, j = w - a.st_off[s];
        const float v = j < He ? lds[a.buf_a[s] + pad4(He) + j] : lds[a.buf_b[s] + pad4(2 * M - 1) + (j - He)];
        st[size_t(w) * lanes + lane] = __float_as_uint(v);
    }
}

// -------------------------------------------------------------- interpolator
// One `HbfInt` stage (src/hbf.rs:207-227) for n inputs -> n pairs:
//   [get(X)[i], X[M + i]]  (interpolated sample, then the centre-tap identity).
template <int M>
__device__ __forceinline__ void int_stage(const float *X, int n, const float (&taps)[IDSP_HBF_MAX_TAPS], float *Xn,
                                          float *yg, bool yvec, int tid)
{
    constexpr int dx = pad4(2 * M - 1);
    constexpr int WX = up4(dx + 2 * M + 3);
    const int nq = (n + 3) >> 2;
    for (int g = tid; g < nq; g += kThreads) {
        const int i0 = g * 4;
        float w[WX];
#pragma unroll
        for (int v = 0; v < WX / 4; v++) {
            const v4f t = *reinterpret_cast<const v4f *>(X + i0 + 4 * v);
            w[4 * v] = t.x, w[4 * v + 1] = t.y, w[4 * v + 2] = t.z, w[4 * v + 3] = t.w;
        }
        float out[8];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            out[2 * p] = window_sum<M>(w, dx + p, taps);
            out[2 * p + 1] = w[dx + M + p];
        }
        if (yg) {
            if (yvec && i0 + 4 <= n) {
                *reinterpret_cast<v4f *>(yg + 2 * i0) = v4f{out[0], out[1], out[2], out[3]};
                *reinterpret_cast<v4f *>(yg + 2 * i0 + 4) = v4f{out[4], out[5], out[6], out[7]};
            } else {
#pragma unroll
                for (int p = 0; p < 8; p++)
                    if (2 * i0 + p < 2 * n) yg[2 * i0 + p] = out[p];
            }
        } else {
            // `ChunkOut<_, 2>`: the pairs flatten into the next stage's input stream
            *reinterpret_cast<v4f *>(Xn + 2 * i0) = v4f{out[0], out[1], out[2], out[3]};
            *reinterpret_cast<v4f *>(Xn + 2 * i0 + 4) = v4f{out[4], out[5], out[6], out[7]};
        }
    }
}

__device__ void int_stage_any(int M, const float *X, int n, const float *taps, float *Xn, float *yg, int tid)
{
    const int dx = pad4(2 * M - 1);
    for (int i = tid; i < n; i += kThreads) {
        float acc = -0.0f;
        for (int k = 0; k < M; k++) acc = acc + (X[dx + i + 2 * M - 1 - k] + X[dx + i + k]) * taps[k];
        float *dst = yg ? yg : Xn;
        dst[2 * i] = acc;
        dst[2 * i + 1] = X[dx + M + i];
    }
}

__device__ __forceinline__ void int_stage_dispatch(int M, const float *X, int n, const float (&taps)[IDSP_HBF_MAX_TAPS],
                                                   float *Xn, float *yg, bool yvec, int tid)
{
    switch (M) {
        case 2: int_stage<2>(X, n, taps, Xn, yg, yvec, tid); break;
        case 3: int_stage<3>(X, n, taps, Xn, yg, yvec, tid); break;
        case 4: int_stage<4>(X, n, taps, Xn, yg, yvec, tid); break;
        case 5: int_stage<5>(X, n, taps, Xn, yg, yvec, tid); break;
        case 6: int_stage<6>(X, n, taps, Xn, yg, yvec, tid); break;
        case 10: int_stage<10>(X, n, taps, Xn, yg, yvec, tid); break;
        case 15: int_stage<15>(X, n, taps, Xn, yg, yvec, tid); break;
        case 23: int_stage<23>(X, n, taps, Xn, yg, yvec, tid); break;
        default: int_stage_any(M, X, n, taps, Xn, yg, tid); break;
    }
}

__global__ __launch_bounds__(kThreads) void hbf_int_kernel(const HbfArgs a, uint32_t *st, const float *x, float *y,
                                                           const size_t lanes, const size_t frames, const int lane_major)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    // FRAME_MAJOR: neighbouring lanes share cache lines -> give each XCD (own L2) a contiguous eighth of the lanes
    const size_t lane = lane_major ? size_t(blockIdx.x) : (blockIdx.x % 8) * ((lanes + 7) / 8) + blockIdx.x / 8;
    if (lane >= lanes) return;
    const int S = a.stages;
    const int R = 1 << S;

    for (int s = 0; s < S; s++) {
        const int H = 2 * a.m[s] - 1;
        float *X = lds + a.buf_a[s] + pad4(H);
        for (int w = tid; w < H; w += kThreads) X[w] = __uint_as_float(st[size_t(a.st_off[s] + w) * lanes + lane]);
    }

    const size_t ch = size_t(kChunk >> S);  // input frames per chunk
    float *X0n = lds + a.buf_a[0] + up4(2 * a.m[0] - 1);
    float *ystage = lds + a.stage_off;  // FRAME_MAJOR output chunks

    // next chunk's inputs are fetched one chunk ahead (ch <= 2048 frames -> <= 8 per thread)
    constexpr int kPreI = (kChunk / 2) / kThreads;
    float pre[kPreI];
    auto fetch = [&](size_t f0) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch);
#pragma unroll
        for (int i = 0; i < kPreI; i++) {
            const int j = tid + i * kThreads;
            if (j < nf) pre[i] = lane_major ? x[lane * frames + f0 + size_t(j)] : x[(f0 + size_t(j)) * lanes + lane];
        }
    };
    fetch(0);

    for (size_t f0 = 0; f0 < frames; f0 += ch) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch);
#pragma unroll
        for (int i = 0; i < kPreI; i++) {
            const int j = tid + i * kThreads;
            if (j < nf) X0n[j] = pre[i];
        }
        if (f0 + ch < frames) fetch(f0 + ch);
        lds_barrier();

        int n = nf;
        for (int s = 0; s < S; s++) {
            const int M = a.m[s];
            const float *X = lds + a.buf_a[s];
            if (s + 1 < S) {
                int_stage_dispatch(M, X, n, a.taps[s], lds + a.buf_a[s + 1] + up4(2 * a.m[s + 1] - 1), nullptr, false, tid);
            } else if (lane_major) {
                float *yg = y + (lane * frames + f0) * size_t(R);
                const bool yvec = reinterpret_cast<uintptr_t>(yg) % 16 == 0;
                int_stage_dispatch(M, X, n, a.taps[s], nullptr, yg, yvec, tid);
            } else {
                int_stage_dispatch(M, X, n, a.taps[s], ystage, nullptr, false, tid);
            }
            lds_barrier();
            n <<= 1;
        }
        if (!lane_major) {
            for (int i = tid; i < nf * R; i += kThreads)
                y[((f0 + size_t(i / R)) * lanes + lane) * size_t(R) + size_t(i % R)] = ystage[i];
            lds_barrier();
        }

        // roll histories: word j <- word n_s + j (src/hbf.rs:224)
        float keep[2];
        int c = 0;
        for (int w = tid; w < a.st_off[S]; w += kThreads, c++) {
            int s = 0;
            while (w >= a.st_off[s + 1]) s++;
            const int H = 2 * a.m[s] - 1, j = w - a.st_off[s];
            keep[c & 1] = lds[a.buf_a[s] + pad4(H) + (nf << s) + j];
        }
        lds_barrier();
        c = 0;
        for (int w = tid; w < a.st_off[S]; w += kThreads, c++) {
            int s = 0;
            while (w >= a.st_off[s + 1]) s++;
            const int H = 2 * a.m[s] - 1, j = w - a.st_off[s];
            lds[a.buf_a[s] + pad4(H) + j] = keep[c & 1];
        }
        lds_barrier();
    }

    for (int w = tid; w < a.st_off[S]; w += kThreads) {
        int s = 0;
        while (w >= a.st_off[s + 1]) s++;
        const int H = 2 * a.m[s] - 1, j = w - a.st_off[s];
        st[size_t(w) * lanes + lane] = __float_as_uint(lds[a.buf_a[s] + pad4(H) + j]);
    }
}

// ------------------------------------------------------ same-rate symmetric FIR
// `type_fir!` (src/hbf.rs:70-138): one 256-thread workgroup per lane, kChunk samples per
// chunk in LDS behind the LEN-sample history; thread i produces output i (stride-1 LDS reads).
struct FirArgs {
    int32_t m, odd, sym;
    float taps[IDSP_HBF_MAX_TAPS];
};

__global__ __launch_bounds__(kThreads) void fir_sym_kernel(const FirArgs a, uint32_t *st, const float *x, float *y,
                                                           const size_t lanes, const size_t frames, const int lane_major)
{
    __shared__ float buf[2 * IDSP_HBF_MAX_TAPS + kChunk + 8];
    const int tid = threadIdx.x;
    // FRAME_MAJOR: neighbouring lanes share cache lines -> give each XCD (own L2) a contiguous eighth of the lanes
    const size_t lane = lane_major ? size_t(blockIdx.x) : (blockIdx.x % 8) * ((lanes + 7) / 8) + blockIdx.x / 8;
    if (lane >= lanes) return;
    const int M = a.m, len = 2 * M - 1 + a.odd, win = 2 * M + a.odd;
    for (int w = tid; w < len; w += kThreads) buf[w] = __uint_as_float(st[size_t(w) * lanes + lane]);
    for (size_t f0 = 0; f0 < frames; f0 += kChunk) {
        const int n = int(frames - f0 < size_t(kChunk) ? frames - f0 : size_t(kChunk));
        for (int i = tid; i < n; i += kThreads)
            buf[len + i] = lane_major ? x[lane * frames + f0 + size_t(i)] : x[(f0 + size_t(i)) * lanes + lane];
        lds_barrier();
        float out[kChunk / kThreads];
#pragma unroll
        for (int j = 0; j < kChunk / kThreads; j++) {
            const int i = tid + j * kThreads;
            if (i < n) {
                const float *w = buf + i;
                float acc = -0.0f;  // f32::sum
                for (int k = 0; k < M; k++) {
                    const float nw = w[win - 1 - k], od = w[k];
                    acc = acc + (a.sym ? nw + od : nw - od) * a.taps[k];
                }
                out[j] = (a.odd && a.sym) ? acc + w[M] : acc;
            }
        }
        float keep = tid < len ? buf[n + tid] : 0.f;  // len <= 64 < kThreads
        lds_barrier();
        if (tid < len) buf[tid] = keep;
#pragma unroll
        for (int j = 0; j < kChunk / kThreads; j++) {
            const int i = tid + j * kThreads;
            if (i < n) {
                if (lane_major)
                    y[lane * frames + f0 + size_t(i)] = out[j];
                else
                    y[(f0 + size_t(i)) * lanes + lane] = out[j];
            }
        }
        lds_barrier();
    }
    for (int w = tid; w < len; w += kThreads) st[size_t(w) * lanes + lane] = __float_as_uint(buf[w]);
}

// ------------------------------------------------------------------- host
int fill_args(const idsp_hbf_cascade_f32 *cfg, bool dec, HbfArgs &a, int &lds_words)
{
    std::memset(&a, 0, sizeof(a));
    a.stages = cfg->stages;
    int off = 0, so = 0;
    for (int s = 0; s < cfg->stages; s++) {
        const int M = cfg->m[s];
        a.m[s] = M;
        for (int k = 0; k < M; k++) a.taps[s][k] = cfg->taps[s][k];
        a.st_off[s] = so;
        if (dec) {
            const int n = kChunk >> (s + 1);  // outputs of stage s per chunk = samples per stream
            a.buf_a[s] = off;
            off += up4(up4(M - 1) + n + kSlack);
            a.buf_b[s] = off;
            off += up4(up4(2 * M - 1) + n + kSlack + 2 * M);
            so += 3 * M - 2;
        } else {
            const int n = (kChunk >> cfg->stages) << s;  // inputs of stage s per chunk
            a.buf_a[s] = off;
            off += up4(up4(2 * M - 1) + n + kSlack + 2 * M);
            so += 2 * M - 1;
        }
    }
    a.st_off[cfg->stages] = so;
#ifdef IDSP_DEBUG_ABLATE  // timing ablation: never in the shipped library (it changes results)
    if (const char *e = diag_env("IDSP_HBF_ABLATE")) a.ablate = atoi(e);
#endif
    a.stage_off = off;  // FRAME_MAJOR output staging: one chunk of outputs
    off += (dec ? kChunk / 2 : kChunk) + kSlack;
    lds_words = off;
    return IDSP_OK;
}

// Which built-in tap set (0 = HBF_TAPS, 1 = HBF_TAPS_98) `cfg` is, bit for bit, or -1.
int builtin_tap_set(const idsp_hbf_cascade_f32 *cfg, bool dec)
{
    for (int ts = 0; ts < 2; ts++) {
        bool same = true;
        for (int s = 0; s < cfg->stages && same; s++) {
            const int t = hbf_tuple_index(dec, cfg->stages, s);
            same = cfg->m[s] == kHbfM[ts][t] && std::memcmp(cfg->taps[s], kHbfTaps[ts][t], sizeof(float) * size_t(cfg->m[s])) == 0;
        }
        if (same) return ts;
    }
    return -1;
}

template <class K>
int launch_hbf(K kernel, const idsp_hbf_cascade_f32 *cfg, bool dec, void *state, const float *x, float *y, size_t lanes,
               size_t frames, int layout, void *stream)
{
    if (!hbf_cfg_ok(cfg)) return fail(IDSP_EINVAL, "invalid hbf cascade (stages 1..5, taps 1..32 per stage)");
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && (!x || !y)))) return fail(IDSP_EINVAL, "state, x or y is NULL");
    if (lanes > (size_t(1) << 31) - 1 || frames > (size_t(1) << 40)) return fail(IDSP_EINVAL, "lanes/frames out of range");
    if (lanes == 0 || frames == 0) return IDSP_OK;
    // f32 slices are 4-byte aligned in the reference; the generic kernels' 8- and 16-byte global accesses are legal at that
    // alignment on gfx950 (unaligned access mode), the wave kernels below are taken only when their accesses are aligned
    // Fast path: the reference's own cascades run on the specialised one-wave-per-lane
    // kernels when every 16-byte access they make is aligned; anything else (custom taps,
    // odd shapes) takes the generic workgroup-per-lane kernel below.
    static const bool force_generic = diag_env("IDSP_HBF_GENERIC") != nullptr;
    const int ts = force_generic ? -1 : builtin_tap_set(cfg, dec);
    const bool lm = layout == IDSP_LANE_MAJOR;
    const size_t R = size_t(1) << cfg->stages;
    const void *wide = dec ? static_cast<const void *>(x) : static_cast<const void *>(y);
    if (ts >= 0 && reinterpret_cast<uintptr_t>(wide) % 16 == 0 && (lm ? (frames * R) % 4 == 0 : R >= 4)) {
        // decimators: the LDS-DMA ring kernels of hbf_ring.h first (LANE_MAJOR any cascade; FRAME_MAJOR /16 on whole
        // 16-lane groups), then the wave-per-lane kernels of hbf_wave.h
        static const bool no_ring = diag_env("IDSP_HBF_NO_RING") != nullptr, no_blk = diag_env("IDSP_HBF_NO_BLK") != nullptr;
        if (dec && !no_blk) {
            const int rb = hbf_blk_dec(ts, cfg->stages, static_cast<uint32_t *>(state), x, y, lanes, frames, lm, as_stream(stream));
            if (rb == 0) return launch_status();
            if (rb != 1) return IDSP_EHIP;
        }
        if (dec && !no_ring) {
            const int rr = hbf_ring_dec(ts, cfg->stages, static_cast<uint32_t *>(state), x, y, lanes, frames, lm, as_stream(stream));
            if (rr == 0) return launch_status();
            if (rr != 1) return IDSP_EHIP;
        }
        const int rc = dec ? hbf_wave_dec(ts, cfg->stages, static_cast<uint32_t *>(state), x, y, lanes, frames, lm, as_stream(stream))
                           : hbf_wave_int(ts, cfg->stages, static_cast<uint32_t *>(state), x, y, lanes, frames, lm, as_stream(stream));
        if (rc == 0) return launch_status();
    }
    HbfArgs a;
    int lds_words = 0;
    fill_args(cfg, dec, a, lds_words);
    const size_t bytes = size_t(lds_words) * sizeof(float);
    if (bytes > 64 * 1024)
        IDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    note_kernel(dec ? "hbf_dec_kernel (generic taps)" : "hbf_int_kernel (generic taps)");
    hipLaunchKernelGGL(kernel, dim3(unsigned(layout == IDSP_LANE_MAJOR ? lanes : 8 * ((lanes + 7) / 8))), dim3(kThreads), bytes, as_stream(stream), a,
                       static_cast<uint32_t *>(state), x, y, lanes, frames, layout == IDSP_LANE_MAJOR ? 1 : 0);
    return launch_status();
}

}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

int idsp_hbf_dec_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y, size_t lanes,
                     size_t frames, int layout, void *stream)
{
    return launch_hbf(hbf_dec_kernel, cfg, true, state, x, y, lanes, frames, layout, stream);
}

int idsp_hbf_int_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y, size_t lanes,
                     size_t frames, int layout, void *stream)
{
    return launch_hbf(hbf_int_kernel, cfg, false, state, x, y, lanes, frames, layout, stream);
}

size_t idsp_fir_sym_state_words(const idsp_fir_sym_f32 *cfg)
{
    if (!cfg || cfg->kind < 0 || cfg->kind > 3 || cfg->m < 1 || cfg->m > IDSP_HBF_MAX_TAPS) return 0;
    return size_t(2 * cfg->m - 1 + (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_ODD_ANTISYMMETRIC ? 1 : 0));
}

int idsp_fir_sym_f32_process(const idsp_fir_sym_f32 *cfg, void *state, const float *x, float *y, size_t lanes,
                             size_t frames, int layout, void *stream)
{
    if (!idsp_fir_sym_state_words(cfg)) return fail(IDSP_EINVAL, "invalid FIR configuration (kind 0..3, m 1..32)");
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && (!x || !y)))) return fail(IDSP_EINVAL, "state, x or y is NULL");
    if (lanes > (size_t(1) << 31) - 1) return fail(IDSP_EINVAL, "lanes out of range");
    if (lanes == 0 || frames == 0) return IDSP_OK;
    FirArgs a;
    a.m = cfg->m;
    a.odd = (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_ODD_ANTISYMMETRIC) ? 1 : 0;
    a.sym = (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_EVEN_SYMMETRIC) ? 1 : 0;
    for (int k = 0; k < IDSP_HBF_MAX_TAPS; k++) a.taps[k] = k < cfg->m ? cfg->taps[k] : 0.f;
    note_kernel("fir_sym_kernel");
    hipLaunchKernelGGL(fir_sym_kernel, dim3(unsigned(layout == IDSP_LANE_MAJOR ? lanes : 8 * ((lanes + 7) / 8))), dim3(kThreads), 0, as_stream(stream), a,
                       static_cast<uint32_t *>(state), x, y, lanes, frames, layout == IDSP_LANE_MAJOR ? 1 : 0);
    return launch_status();
}

}  // extern "C"
