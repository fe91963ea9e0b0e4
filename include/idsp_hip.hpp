// idsp_hip.hpp — header-only C++17 host layer above the C ABI (idsp_hip.h).
//
// The reference is compiled code (Rust) and no Rust toolchain exists in the
// build image, so the host side that mirrors its operator interface is C++:
// the same names, argument meaning and error behaviour as `dsp_process` /
// `idsp`, resolved statically like the reference's traits.  Needs only the C
// ABI (no HIP headers): buffers are owned through idsp_device_alloc.
//
//   reference                                                   here
//   ----------------------------------------------------------  --------------------------------
//   Biquad<Q32<F>> / Biquad<f32>        (src/iir/biquad.rs:96)  Biquad<Q32<F>> / Biquad<float>
//   BiquadClamp<C, T>                   (src/iir/biquad.rs:121) BiquadClamp<C>
//   DirectForm1<T> / DirectForm2Transposed<T> / DirectForm1Wide / DirectForm1Dither   (state tags)
//   Split::new(cfg, S::default()).lanes::<N>()  (split.rs:272)  Split(cfg, S{}).lanes(n)
//   Process::block / Inplace::inplace           (process.rs:34) .block(x, y) / .inplace(xy)   FrameMajor
//   ViewProcess::process_view(View<LaneMajor>)  (view.rs:245)   .process_view(View, ViewMut)
//   HBF_DEC_CASCADE + HbfDec2..32 / HBF_INT_CASCADE (hbf.rs)    HbfDecCascade / HbfIntCascade
//   Lockin<[Lowpass<N>; K]> + Accu      (lockin.rs, accu.rs)    Lockin<N, K>
//   ByLane<[C; N]>                      (compose.rs:363)        ByLane<Cfg, S>(configs)
//   coefficients::Filter, pid::Builder, Pid, Units (iir/coefficients.rs, iir/pid.rs)  Filter, pid::Builder, Pid, Units
//   Split::stateful(Cic::new(rate)).decimate()/.interpolate() (cic.rs:338-346)  CicDecimator<T> / CicInterpolator<T>
//   cossin(phase)                       (cossin.rs:14)          cossin(phases, out)
//   atan2(y, x) / Complex::arg          (atan2.rs:66)           atan2(xy, out)
//
// Misuse that is a `debug_assert!`/panic in the reference throws idsp_hip::Error.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "idsp_hip.h"

namespace idsp_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
    if (rc < 0) throw Error(rc, std::string("idsp status ") + std::to_string(rc) + ": " + idsp_last_error());
}
inline void require(bool ok, const char *what)
{
    if (!ok) throw Error(IDSP_EINVAL, what);
}

// ------------------------------------------------------------------ memory
/// Device-resident slice `[T]` (the engine never allocates behind the caller's back).
template <class T>
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n, bool zero = true) : n_(n)
    {
        if (n) {
            check(idsp_device_alloc(reinterpret_cast<void **>(&p_), n * sizeof(T)));
            if (zero) check(idsp_device_memset(p_, 0, n * sizeof(T), nullptr));
        }
    }
    explicit DeviceBuffer(const std::vector<T> &host) : DeviceBuffer(host.size(), false) { upload(host); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr, o.n_ = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
    {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        return *this;
    }
    ~DeviceBuffer()
    {
        if (p_) idsp_device_free(p_);
    }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t len() const { return n_; }
    void upload(const std::vector<T> &host)
    {
        require(host.size() == n_, "upload: length mismatch");
        if (n_) {
            check(idsp_device_h2d(p_, host.data(), n_ * sizeof(T), nullptr));
            check(idsp_stream_sync(nullptr));
        }
    }
    std::vector<T> to_host() const
    {
        std::vector<T> h(n_);
        if (n_) {
            check(idsp_device_d2h(h.data(), p_, n_ * sizeof(T), nullptr));
            check(idsp_stream_sync(nullptr));
        }
        return h;
    }

private:
    T *p_ = nullptr;
    size_t n_ = 0;
};

// ------------------------------------------------------------------- views
struct FrameMajor {  // dsp-process/src/view.rs:10
    static constexpr int value = IDSP_FRAME_MAJOR;
};
struct LaneMajor {  // dsp-process/src/view.rs:17
    static constexpr int value = IDSP_LANE_MAJOR;
};

/// `View<'_, T, Layout, L>` with a runtime lane count (dsp-process/src/view.rs:24-28).
template <class T, class Layout>
struct View {
    const T *flat;
    size_t frames, lanes;
    /// `from_flat`: `assert_eq!(flat.len(), frames * L)` (view.rs:182)
    static View from_flat(const DeviceBuffer<T> &b, size_t lanes, size_t width = 1)
    {
        require(lanes && b.len() % (lanes * width) == 0, "flat.len() is not frames * L");
        return View{b.data(), b.len() / (lanes * width), lanes};
    }
};
template <class T, class Layout>
struct ViewMut {
    T *flat;
    size_t frames, lanes;
    static ViewMut from_flat(DeviceBuffer<T> &b, size_t lanes, size_t width = 1)
    {
        require(lanes && b.len() % (lanes * width) == 0, "flat.len() is not frames * L");
        return ViewMut{b.data(), b.len() / (lanes * width), lanes};
    }
    operator View<T, Layout>() const { return View<T, Layout>{flat, frames, lanes}; }
};

// ---------------------------------------------------------- configurations
/// Coefficient type tag `Q32<F>` = `Q<i32, i64, F>` (dsp-fixedpoint/src/lib.rs:474-492).
template <int F>
struct Q32 {
    static_assert(F >= 0 && F < 32, "0 <= F < 32");
    static constexpr int frac = F;
};

template <class C>
struct Biquad;  // src/iir/biquad.rs:96-116: ba = [b0, b1, b2, a1, a2], a1/a2 as used

template <int F>
struct Biquad<Q32<F>> {
    using Sample = int32_t;
    std::array<int32_t, 5> ba{};  // raw Q bits
    /// `Biquad::from([[b0,b1,b2],[a0,a1,a2]])` (biquad.rs:545-566) + f64 -> Q32<F> (num_traits_impl.rs:32-46)
    static Biquad from_sos(const std::array<double, 6> &sos)
    {
        idsp_biquad_i32 q;
        check(idsp_biquad_i32_from_sos(sos.data(), F, &q));
        Biquad b;
        std::memcpy(b.ba.data(), q.ba, sizeof(q.ba));
        return b;
    }
    static Biquad proportional(int32_t k) { return Biquad{{k, 0, 0, 0, 0}}; }          // biquad.rs:196-200
    static Biquad identity() { return proportional(int32_t(1) << F); }                   // biquad.rs:184
    static Biquad hold() { return Biquad{{0, 0, 0, int32_t(1) << F, 0}}; }               // biquad.rs:212-214
    int32_t forward_gain() const { return int32_t(uint32_t(ba[0]) + uint32_t(ba[1]) + uint32_t(ba[2])); }
    idsp_biquad_i32 abi() const
    {
        idsp_biquad_i32 q;
        std::memcpy(q.ba, ba.data(), sizeof(q.ba));
        q.frac = F;
        return q;
    }
};

template <>
struct Biquad<float> {
    using Sample = float;
    std::array<float, 5> ba{};
    static Biquad from_sos(const std::array<float, 6> &sos)
    {
        idsp_biquad_f32 q;
        check(idsp_biquad_f32_from_sos(sos.data(), &q));
        Biquad b;
        std::memcpy(b.ba.data(), q.ba, sizeof(q.ba));
        return b;
    }
    static Biquad from_sos(const std::array<double, 6> &sos)
    {
        idsp_biquad_f32 q;
        check(idsp_biquad_f32_from_sos_f64(sos.data(), &q));
        Biquad b;
        std::memcpy(b.ba.data(), q.ba, sizeof(q.ba));
        return b;
    }
    static Biquad proportional(float k) { return Biquad{{k, 0, 0, 0, 0}}; }
    static Biquad identity() { return proportional(1.0f); }
    static Biquad hold() { return Biquad{{0, 0, 0, 1.0f, 0}}; }
    float forward_gain() const { return ba[0] + ba[1] + ba[2]; }
    idsp_biquad_f32 abi() const
    {
        idsp_biquad_f32 q;
        std::memcpy(q.ba, ba.data(), sizeof(q.ba));
        return q;
    }
};

template <>
struct Biquad<double> {
    using Sample = double;
    std::array<double, 5> ba{};
    static Biquad from_sos(const std::array<double, 6> &sos)
    {
        idsp_biquad_f64 q;
        check(idsp_biquad_f64_from_sos(sos.data(), &q));
        Biquad b;
        std::memcpy(b.ba.data(), q.ba, sizeof(q.ba));
        return b;
    }
    static Biquad proportional(double k) { return Biquad{{k, 0, 0, 0, 0}}; }
    static Biquad identity() { return proportional(1.0); }
    static Biquad hold() { return Biquad{{0, 0, 0, 1.0, 0}}; }
    double forward_gain() const { return ba[0] + ba[1] + ba[2]; }
    idsp_biquad_f64 abi() const
    {
        idsp_biquad_f64 q;
        std::memcpy(q.ba, ba.data(), sizeof(q.ba));
        return q;
    }
};

/// `BiquadClamp<C, T>` (biquad.rs:121-171); defaults u = 0, min = T::MIN, max = T::MAX
/// (+-inf for floats, src/num.rs:33-52).
template <class C>
struct BiquadClamp {
    using Sample = typename Biquad<C>::Sample;
    Biquad<C> coeff{};
    Sample u = Sample(0);
    Sample min = std::numeric_limits<Sample>::has_infinity ? -std::numeric_limits<Sample>::infinity()
                                                           : std::numeric_limits<Sample>::lowest();
    Sample max = std::numeric_limits<Sample>::has_infinity ? std::numeric_limits<Sample>::infinity()
                                                           : std::numeric_limits<Sample>::max();
    BiquadClamp() = default;
    BiquadClamp(const Biquad<C> &c) : coeff(c) {}  // `From<F: Into<Biquad<C>>>` (biquad.rs:578-588)
};

// state kinds (tags): words per lane and section
// words are per 32-bit value; f64 states take twice as many (Lanes multiplies by sizeof(Sample)/4)
struct DirectForm1 { static constexpr int words = 4; };            // biquad.rs:319
struct DirectForm2Transposed { static constexpr int words = 2; };  // biquad.rs:407
struct DirectForm1Wide { static constexpr int words = 6; };        // biquad.rs:445-454
struct DirectForm1Dither { static constexpr int words = 5; };      // biquad.rs:484-491

namespace detail {
using StreamI32 = int (*)(const idsp_biquad_i32 *, size_t, void *, const int32_t *, int32_t *, size_t, size_t, int, void *);
using StreamClampI32 = int (*)(const idsp_biquad_clamp_i32 *, size_t, void *, const int32_t *, int32_t *, size_t, size_t, int, void *);
using StreamF32 = int (*)(const idsp_biquad_f32 *, size_t, void *, const float *, float *, size_t, size_t, int, void *);
using StreamClampF32 = int (*)(const idsp_biquad_clamp_f32 *, size_t, void *, const float *, float *, size_t, size_t, int, void *);
using StreamF64 = int (*)(const idsp_biquad_f64 *, size_t, void *, const double *, double *, size_t, size_t, int, void *);
using StreamClampF64 = int (*)(const idsp_biquad_clamp_f64 *, size_t, void *, const double *, double *, size_t, size_t, int, void *);

// (configuration, state) -> C entry point; a missing specialisation is the reference's
// "trait not implemented" compile error.
template <class Cfg, class S>
struct Entry;
template <int F> struct Entry<Biquad<Q32<F>>, DirectForm1> { static constexpr StreamI32 fn = idsp_biquad_i32_df1; };
template <int F> struct Entry<Biquad<Q32<F>>, DirectForm1Dither> { static constexpr StreamI32 fn = idsp_biquad_i32_dither; };
template <int F> struct Entry<Biquad<Q32<F>>, DirectForm1Wide> { static constexpr StreamI32 fn = idsp_biquad_i32_wide; };
template <int F> struct Entry<BiquadClamp<Q32<F>>, DirectForm1> { static constexpr StreamClampI32 fn = idsp_biquad_i32_df1_clamp; };
template <int F> struct Entry<BiquadClamp<Q32<F>>, DirectForm1Dither> { static constexpr StreamClampI32 fn = idsp_biquad_i32_dither_clamp; };
template <int F> struct Entry<BiquadClamp<Q32<F>>, DirectForm1Wide> { static constexpr StreamClampI32 fn = idsp_biquad_i32_wide_clamp; };
template <> struct Entry<Biquad<float>, DirectForm1> { static constexpr StreamF32 fn = idsp_biquad_f32_df1; };
template <> struct Entry<Biquad<float>, DirectForm2Transposed> { static constexpr StreamF32 fn = idsp_biquad_f32_df2t; };
template <> struct Entry<BiquadClamp<float>, DirectForm1> { static constexpr StreamClampF32 fn = idsp_biquad_f32_df1_clamp; };
template <> struct Entry<BiquadClamp<float>, DirectForm2Transposed> { static constexpr StreamClampF32 fn = idsp_biquad_f32_df2t_clamp; };

template <> struct Entry<Biquad<double>, DirectForm1> { static constexpr StreamF64 fn = idsp_biquad_f64_df1; };
template <> struct Entry<Biquad<double>, DirectForm2Transposed> { static constexpr StreamF64 fn = idsp_biquad_f64_df2t; };
template <> struct Entry<BiquadClamp<double>, DirectForm1> { static constexpr StreamClampF64 fn = idsp_biquad_f64_df1_clamp; };
template <> struct Entry<BiquadClamp<double>, DirectForm2Transposed> { static constexpr StreamClampF64 fn = idsp_biquad_f64_df2t_clamp; };

inline idsp_biquad_f64 to_abi(const Biquad<double> &b) { return b.abi(); }
inline idsp_biquad_clamp_f64 to_abi(const BiquadClamp<double> &c)
{
    idsp_biquad_clamp_f64 q;
    std::memcpy(q.ba, c.coeff.ba.data(), sizeof(q.ba));
    q.u = c.u, q.min = c.min, q.max = c.max;
    return q;
}
template <int F> idsp_biquad_i32 to_abi(const Biquad<Q32<F>> &b) { return b.abi(); }
inline idsp_biquad_f32 to_abi(const Biquad<float> &b) { return b.abi(); }
template <int F>
idsp_biquad_clamp_i32 to_abi(const BiquadClamp<Q32<F>> &c)
{
    idsp_biquad_clamp_i32 q;
    std::memcpy(q.ba, c.coeff.ba.data(), sizeof(q.ba));
    q.frac = F, q.u = c.u, q.min = c.min, q.max = c.max;
    return q;
}
inline idsp_biquad_clamp_f32 to_abi(const BiquadClamp<float> &c)
{
    idsp_biquad_clamp_f32 q;
    std::memcpy(q.ba, c.coeff.ba.data(), sizeof(q.ba));
    q.u = c.u, q.min = c.min, q.max = c.max;
    return q;
}
}  // namespace detail

/// `Split<Lanes<C>, [S; lanes]>` resident on the GPU: one shared configuration (or a serial
/// slice of sections, `[C] x [S]`, dsp-process/src/compose.rs:43-77), `lanes` independent states.
template <class Cfg, class S>
class Lanes {
public:
    using Sample = typename Cfg::Sample;
    using Abi = decltype(detail::to_abi(std::declval<Cfg>()));

    Lanes(const std::vector<Cfg> &sections, size_t lanes, void *stream = nullptr)
        : lanes_(lanes), stream_(stream),
          state_(size_t(S::words) * (sizeof(Sample) / 4) * (sections.empty() ? 1 : sections.size()) * lanes)
    {
        for (const auto &c : sections) abi_.push_back(detail::to_abi(c));
    }
    size_t lanes() const { return lanes_; }
    /// per-lane state words, word-plane-major; zero == `Default::default()`
    DeviceBuffer<uint32_t> &state() { return state_; }

    /// `Process::block(&mut self, x: &[[X; N]], y: &mut [[Y; N]])` — FrameMajor (process.rs:44-49)
    void block(const DeviceBuffer<Sample> &x, DeviceBuffer<Sample> &y)
    {
        require(x.len() == y.len(), "x.len() != y.len()");  // process.rs:45 debug_assert_eq!
        require(lanes_ && x.len() % lanes_ == 0, "slice is not a whole number of frames");
        run(x.data(), y.data(), x.len() / lanes_, IDSP_FRAME_MAJOR);
    }
    /// `Inplace::inplace(&mut self, xy: &mut [[X; N]])` (process.rs:61-65)
    void inplace(DeviceBuffer<Sample> &xy)
    {
        require(lanes_ && xy.len() % lanes_ == 0, "slice is not a whole number of frames");
        run(xy.data(), xy.data(), xy.len() / lanes_, IDSP_FRAME_MAJOR);
    }
    /// `ViewProcess::process_view` (view.rs:245-248; `Lanes` lane-slice path compose.rs:478-494)
    template <class Layout>
    void process_view(View<Sample, Layout> x, ViewMut<Sample, Layout> y)
    {
        require(x.frames == y.frames, "x.frames() != y.frames()");  // compose.rs:488
        require(x.lanes == lanes_ && y.lanes == lanes_, "view lane count != Lanes lane count");
        run(x.flat, y.flat, x.frames, Layout::value);
    }
    template <class Layout>
    void inplace_view(ViewMut<Sample, Layout> xy)
    {
        require(xy.lanes == lanes_, "view lane count != Lanes lane count");
        run(xy.flat, xy.flat, xy.frames, Layout::value);
    }

private:
    void run(const Sample *x, Sample *y, size_t frames, int layout)
    {
        check(detail::Entry<Cfg, S>::fn(abi_.data(), abi_.size(), state_.data(), x, y, lanes_, frames, layout, stream_));
    }
    size_t lanes_;
    void *stream_;
    std::vector<Abi> abi_;
    DeviceBuffer<uint32_t> state_;
};

/// `Split<C, S>` (dsp-process/src/split.rs:29-34).
template <class Cfg, class S>
struct SplitT {
    std::vector<Cfg> sections;
    /// `.lanes::<N>()` (split.rs:272-277)
    Lanes<Cfg, S> lanes(size_t n, void *stream = nullptr) const { return Lanes<Cfg, S>(sections, n, stream); }
};
template <class Cfg, class S>
SplitT<Cfg, S> Split(const Cfg &cfg, S)
{
    return SplitT<Cfg, S>{{cfg}};
}
template <class Cfg, class S>
SplitT<Cfg, S> Split(const std::vector<Cfg> &sections, S)
{
    return SplitT<Cfg, S>{sections};
}

namespace detail {
using ByLaneI32 = int (*)(const int32_t *, int, size_t, void *, const int32_t *, int32_t *, size_t, size_t, int, void *);
using ByLaneF32 = int (*)(const float *, size_t, void *, const float *, float *, size_t, size_t, int, void *);
using ByLaneF64 = int (*)(const double *, size_t, void *, const double *, double *, size_t, size_t, int, void *);
template <class Cfg, class S>
struct EntryByLane;
template <int F> struct EntryByLane<Biquad<Q32<F>>, DirectForm1> { static constexpr ByLaneI32 fn = idsp_biquad_i32_df1_bylane; };
template <int F> struct EntryByLane<Biquad<Q32<F>>, DirectForm1Dither> { static constexpr ByLaneI32 fn = idsp_biquad_i32_dither_bylane; };
template <int F> struct EntryByLane<Biquad<Q32<F>>, DirectForm1Wide> { static constexpr ByLaneI32 fn = idsp_biquad_i32_wide_bylane; };
template <int F> struct EntryByLane<BiquadClamp<Q32<F>>, DirectForm1> { static constexpr ByLaneI32 fn = idsp_biquad_i32_df1_clamp_bylane; };
template <int F> struct EntryByLane<BiquadClamp<Q32<F>>, DirectForm1Dither> { static constexpr ByLaneI32 fn = idsp_biquad_i32_dither_clamp_bylane; };
template <int F> struct EntryByLane<BiquadClamp<Q32<F>>, DirectForm1Wide> { static constexpr ByLaneI32 fn = idsp_biquad_i32_wide_clamp_bylane; };
template <> struct EntryByLane<Biquad<float>, DirectForm1> { static constexpr ByLaneF32 fn = idsp_biquad_f32_df1_bylane; };
template <> struct EntryByLane<Biquad<float>, DirectForm2Transposed> { static constexpr ByLaneF32 fn = idsp_biquad_f32_df2t_bylane; };
template <> struct EntryByLane<BiquadClamp<float>, DirectForm1> { static constexpr ByLaneF32 fn = idsp_biquad_f32_df1_clamp_bylane; };
template <> struct EntryByLane<BiquadClamp<float>, DirectForm2Transposed> { static constexpr ByLaneF32 fn = idsp_biquad_f32_df2t_clamp_bylane; };
template <> struct EntryByLane<Biquad<double>, DirectForm1> { static constexpr ByLaneF64 fn = idsp_biquad_f64_df1_bylane; };
template <> struct EntryByLane<Biquad<double>, DirectForm2Transposed> { static constexpr ByLaneF64 fn = idsp_biquad_f64_df2t_bylane; };
template <> struct EntryByLane<BiquadClamp<double>, DirectForm1> { static constexpr ByLaneF64 fn = idsp_biquad_f64_df1_clamp_bylane; };
template <> struct EntryByLane<BiquadClamp<double>, DirectForm2Transposed> { static constexpr ByLaneF64 fn = idsp_biquad_f64_df2t_clamp_bylane; };

template <class C> struct FracOf { static constexpr int value = -1; };
template <int F> struct FracOf<Biquad<Q32<F>>> { static constexpr int value = F; };
template <int F> struct FracOf<BiquadClamp<Q32<F>>> { static constexpr int value = F; };
// one lane's section as the 5 (ba) or 8 (ba, u, min, max) plane values
template <class C> std::vector<typename Biquad<C>::Sample> plane_values(const Biquad<C> &b)
{
    return {b.ba.begin(), b.ba.end()};
}
template <class C> std::vector<typename Biquad<C>::Sample> plane_values(const BiquadClamp<C> &c)
{
    std::vector<typename Biquad<C>::Sample> v(c.coeff.ba.begin(), c.coeff.ba.end());
    v.push_back(c.u), v.push_back(c.min), v.push_back(c.max);
    return v;
}
}  // namespace detail

/// `Split<ByLane<[C; lanes]>, [S; lanes]>` (dsp-process/src/compose.rs:363-390): lane i is filtered by
/// configuration i with state i.  `configs[i]` is the serial slice of sections of lane i; every lane
/// has the same number of sections.  The coefficients are kept on the GPU as lane-contiguous planes.
template <class Cfg, class S>
class ByLane {
public:
    using Sample = typename Cfg::Sample;
    explicit ByLane(const std::vector<std::vector<Cfg>> &configs, void *stream = nullptr)
        : lanes_(configs.size()), sections_(configs.empty() ? 0 : configs[0].size()), stream_(stream),
          state_(size_t(S::words) * (sizeof(Sample) / 4) * (sections_ ? sections_ : 1) * lanes_)
    {
        require(lanes_ && sections_, "ByLane needs at least one lane and one section");
        const size_t cv = detail::plane_values(configs[0][0]).size();
        std::vector<Sample> host(sections_ * cv * lanes_);
        for (size_t l = 0; l < lanes_; l++) {
            require(configs[l].size() == sections_, "every lane needs the same number of sections");
            for (size_t k = 0; k < sections_; k++) {
                const auto v = detail::plane_values(configs[l][k]);
                for (size_t i = 0; i < cv; i++) host[(k * cv + i) * lanes_ + l] = v[i];
            }
        }
        coef_ = DeviceBuffer<Sample>(host);
    }
    /// one section per lane
    explicit ByLane(const std::vector<Cfg> &configs, void *stream = nullptr) : ByLane(wrap(configs), stream) {}
    size_t lanes() const { return lanes_; }
    DeviceBuffer<uint32_t> &state() { return state_; }
    DeviceBuffer<Sample> &coefficients() { return coef_; }

    /// `Process::block` over `&[[X; N]]` — FrameMajor (compose.rs:363-372 per frame)
    void block(const DeviceBuffer<Sample> &x, DeviceBuffer<Sample> &y)
    {
        require(x.len() == y.len(), "x.len() != y.len()");
        require(x.len() % lanes_ == 0, "slice is not a whole number of frames");
        run(x.data(), y.data(), x.len() / lanes_, IDSP_FRAME_MAJOR);
    }
    void inplace(DeviceBuffer<Sample> &xy)
    {
        require(xy.len() % lanes_ == 0, "slice is not a whole number of frames");
        run(xy.data(), xy.data(), xy.len() / lanes_, IDSP_FRAME_MAJOR);
    }
    /// `SplitViewProcess::process_view` (compose.rs:375-389)
    template <class Layout>
    void process_view(View<Sample, Layout> x, ViewMut<Sample, Layout> y)
    {
        require(x.frames == y.frames, "x.frames() != y.frames()");  // compose.rs:386
        require(x.lanes == lanes_ && y.lanes == lanes_, "view lane count != ByLane lane count");
        run(x.flat, y.flat, x.frames, Layout::value);
    }

private:
    static std::vector<std::vector<Cfg>> wrap(const std::vector<Cfg> &c)
    {
        std::vector<std::vector<Cfg>> r;
        for (const auto &v : c) r.push_back({v});
        return r;
    }
    void run(const Sample *x, Sample *y, size_t frames, int layout)
    {
        if constexpr (detail::FracOf<Cfg>::value >= 0)
            check(detail::EntryByLane<Cfg, S>::fn(coef_.data(), detail::FracOf<Cfg>::value, sections_, state_.data(), x, y,
                                                  lanes_, frames, layout, stream_));
        else
            check(detail::EntryByLane<Cfg, S>::fn(coef_.data(), sections_, state_.data(), x, y, lanes_, frames, layout, stream_));
    }
    size_t lanes_, sections_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
    DeviceBuffer<Sample> coef_;
};

// ------------------------------------------------------- coefficient front-end
namespace detail {
template <class C> struct OutOf;
template <int F> struct OutOf<Q32<F>> {
    using Rec = idsp_biquad_clamp_i32;
    static int pid(const idsp_pid_builder *b, double t, int v, int32_t *ba) { return idsp_pid_build_i32(b, t, v, F, ba); }
    static int pid_clamp(const idsp_pid *p, const idsp_units *u, int v, Rec *o) { return idsp_pid_build_clamp_i32(p, u, v, F, o); }
    static int ba(const idsp_ba_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_ba_build_i32(c, u, v, F, o); }
    static int filter(const idsp_filter_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_filter_build_i32(c, u, v, F, o); }
};
template <> struct OutOf<float> {
    using Rec = idsp_biquad_clamp_f32;
    static int pid(const idsp_pid_builder *b, double t, int v, float *ba) { return idsp_pid_build_f32(b, t, v, ba); }
    static int pid_clamp(const idsp_pid *p, const idsp_units *u, int v, Rec *o) { return idsp_pid_build_clamp_f32(p, u, v, o); }
    static int ba(const idsp_ba_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_ba_build_f32(c, u, v, o); }
    static int filter(const idsp_filter_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_filter_build_f32(c, u, v, o); }
};
template <> struct OutOf<double> {
    using Rec = idsp_biquad_clamp_f64;
    static int pid(const idsp_pid_builder *b, double t, int v, double *ba) { return idsp_pid_build_f64(b, t, v, ba); }
    static int pid_clamp(const idsp_pid *p, const idsp_units *u, int v, Rec *o) { return idsp_pid_build_clamp_f64(p, u, v, o); }
    static int ba(const idsp_ba_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_ba_build_f64(c, u, v, o); }
    static int filter(const idsp_filter_config *c, const idsp_units *u, int v, Rec *o) { return idsp_config_filter_build_f64(c, u, v, o); }
};
template <class C, class Rec>
BiquadClamp<C> clamp_from(const Rec &r)
{
    BiquadClamp<C> c;
    for (int i = 0; i < 5; i++) c.coeff.ba[i] = r.ba[i];
    c.u = r.u, c.min = r.min, c.max = r.max;
    return c;
}
}  // namespace detail

/// `pid::Units<T>` (src/iir/pid.rs:350-375)
struct Units : idsp_units {
    Units(double t_ = 1.0, double x_ = 1.0, double y_ = 1.0) : idsp_units{t_, x_, y_} {}
};

/// `coefficients::Filter<T>` (src/iir/coefficients.rs:27-40) with T = f64 (f32 = true: T = f32);
/// setters as in the reference (:110-238).  `build*` is unchecked, `try_build*` validates and throws.
class Filter {
public:
    explicit Filter(bool f32 = false) : f_{0.0, 1.0, 1.0, f32 ? double(1.0f / std::sqrt(2.0f)) : 1.0 / std::sqrt(2.0), IDSP_SHAPE_Q, f32} {}
    Filter &frequency(double critical, double sample) { return critical_frequency(t(t(critical) / t(sample))); }
    Filter &critical_frequency(double f0) { return angular_critical_frequency(t(t(6.283185307179586476925286766559) * t(f0))); }
    Filter &angular_critical_frequency(double w0) { f_.frequency = t(w0); return *this; }
    Filter &gain(double k) { f_.gain = t(k); return *this; }
    Filter &gain_db(double k_db) { return gain(pow10(k_db)); }
    Filter &shelf(double a) { f_.shelf = t(a); return *this; }
    Filter &shelf_db(double a_db) { return shelf(pow10(a_db)); }
    Filter &inverse_q(double qi) { return q(t(1.0 / t(qi))); }
    Filter &q(double v) { return set_shape(IDSP_SHAPE_Q, v); }
    Filter &bandwidth(double bw) { return set_shape(IDSP_SHAPE_BANDWIDTH, bw); }
    Filter &shelf_slope(double s) { return set_shape(IDSP_SHAPE_SLOPE, s); }
    Filter &set_shape(idsp_shape_kind kind, double v) { f_.shape_kind = kind, f_.shape = t(v); return *this; }
    /// `Filter::build(typ)` (:483-495) -> [b0, b1, b2, a0, a1, a2]
    std::array<double, 6> build(idsp_filter_type typ) const { return go(typ, 0); }
    /// `Filter::try_build(typ)` (:498-501)
    std::array<double, 6> try_build(idsp_filter_type typ) const { return go(typ, 1); }
    void validate() const { go(IDSP_LOWPASS, 1); }
    std::array<double, 6> lowpass() const { return build(IDSP_LOWPASS); }
    std::array<double, 6> highpass() const { return build(IDSP_HIGHPASS); }
    std::array<double, 6> bandpass() const { return build(IDSP_BANDPASS); }
    std::array<double, 6> allpass() const { return build(IDSP_ALLPASS); }
    std::array<double, 6> notch() const { return build(IDSP_NOTCH); }
    std::array<double, 6> peaking() const { return build(IDSP_PEAKING); }
    std::array<double, 6> lowshelf() const { return build(IDSP_LOWSHELF); }
    std::array<double, 6> highshelf() const { return build(IDSP_HIGHSHELF); }
    std::array<double, 6> iho() const { return build(IDSP_IHO); }
    /// `build_biquad::<C>` / `try_build_biquad::<C>` (:504-517): `From<[[T; 3]; 2]>` evaluated in T
    template <class C> Biquad<C> build_biquad(idsp_filter_type typ) const { return to_biquad<C>(build(typ)); }
    template <class C> Biquad<C> try_build_biquad(idsp_filter_type typ) const { return to_biquad<C>(try_build(typ)); }
    template <class C> BiquadClamp<C> build_clamped(idsp_filter_type typ) const { return BiquadClamp<C>(build_biquad<C>(typ)); }

private:
    double t(double v) const { return f_.f32 ? double(float(v)) : v; }
    double pow10(double db) const { return f_.f32 ? double(std::pow(10.0f, float(db) / 20.0f)) : std::pow(10.0, db / 20.0); }
    std::array<double, 6> go(idsp_filter_type typ, int validate) const
    {
        std::array<double, 6> ba{};
        check(idsp_filter_build(&f_, typ, validate, ba.data()));
        return ba;
    }
    template <class C>
    Biquad<C> to_biquad(const std::array<double, 6> &ba) const
    {
        idsp_ba_config c{};
        for (int i = 0; i < 6; i++) c.ba[i] = ba[i];
        c.min = -std::numeric_limits<double>::infinity(), c.max = std::numeric_limits<double>::infinity(), c.f32 = f_.f32;
        const Units one;
        typename detail::OutOf<C>::Rec r;
        check(detail::OutOf<C>::ba(&c, &one, 0, &r));
        return detail::clamp_from<C>(r).coeff;
    }
    idsp_filter f_;
};

namespace pid {
enum Action { I2 = 0, I = 1, P = 2, D = 3, D2 = 4 };                  // src/iir/pid.rs:61-75
enum Order { OrderP = 2, OrderI = 1, OrderI2 = 0 };                  // src/iir/pid.rs:14-24

/// `pid::Builder<T>` (src/iir/pid.rs:40-55)
class Builder {
public:
    explicit Builder(bool f32 = false)
    {
        b_.order = OrderI, b_.f32 = f32;
        for (int i = 0; i < 5; i++) b_.gain[i] = 0.0, b_.limit[i] = std::numeric_limits<double>::infinity();
    }
    Builder &order(Order o) { b_.order = o; return *this; }
    Builder &gain(Action a, double g) { b_.gain[a] = g; return *this; }
    Builder &limit(Action a, double l) { b_.limit[a] = l; return *this; }
    Builder &kp(double g) { return gain(P, g); }
    Builder &ki(double g) { return gain(I, g); }
    Builder &ki2(double g) { return gain(I2, g); }
    Builder &kd(double g) { return gain(D, g); }
    Builder &kd2(double g) { return gain(D2, g); }
    Builder &limit_i(double l) { return limit(I, l); }
    Builder &limit_i2(double l) { return limit(I2, l); }
    Builder &limit_d(double l) { return limit(D, l); }
    Builder &limit_d2(double l) { return limit(D2, l); }
    /// `Build<Biquad<C>>::build(&period)` (pid.rs:256-328) / `try_build` (:225-232)
    template <class C> Biquad<C> build(double period) const { return go<C>(period, 0); }
    template <class C> Biquad<C> try_build(double period) const { return go<C>(period, 1); }
    const idsp_pid_builder &abi() const { return b_; }

private:
    template <class C>
    Biquad<C> go(double period, int validate) const
    {
        Biquad<C> out;
        check(detail::OutOf<C>::pid(&b_, period, validate, out.ba.data()));
        return out;
    }
    idsp_pid_builder b_{};
};
}  // namespace pid

/// `pid::Pid<T>` (src/iir/pid.rs:384-431) == `config::PidConfig<T>`
class Pid {
public:
    explicit Pid(bool f32 = false) : b_(f32) {}
    Pid &order(pid::Order o) { b_.order(o); return *this; }
    Pid &kp(double g) { b_.kp(g); return *this; }
    Pid &ki(double g) { b_.ki(g); return *this; }
    Pid &ki2(double g) { b_.ki2(g); return *this; }
    Pid &kd(double g) { b_.kd(g); return *this; }
    Pid &kd2(double g) { b_.kd2(g); return *this; }
    Pid &limit_i(double l) { b_.limit_i(l); return *this; }
    Pid &limit_i2(double l) { b_.limit_i2(l); return *this; }
    Pid &limit_d(double l) { b_.limit_d(l); return *this; }
    Pid &limit_d2(double l) { b_.limit_d2(l); return *this; }
    Pid &setpoint(double s) { setpoint_ = s; return *this; }
    Pid &output_limits(double mn, double mx) { min_ = mn, max_ = mx; return *this; }
    /// `Build<BiquadClamp<C, Y>> for Pid<T>` (pid.rs:533-567) / `try_build` (:521-528)
    template <class C> BiquadClamp<C> build(const Units &u) const { return go<C>(u, 0); }
    template <class C> BiquadClamp<C> try_build(const Units &u) const { return go<C>(u, 1); }

private:
    template <class C>
    BiquadClamp<C> go(const Units &u, int validate) const
    {
        const idsp_pid p{b_.abi(), setpoint_, min_, max_};
        typename detail::OutOf<C>::Rec r;
        check(detail::OutOf<C>::pid_clamp(&p, &u, validate, &r));
        return detail::clamp_from<C>(r);
    }
    pid::Builder b_;
    double setpoint_ = 0.0, min_ = -std::numeric_limits<double>::infinity(), max_ = std::numeric_limits<double>::infinity();
};

/// `BiquadConfig::Ba(BaConfig)` and `BiquadConfig::Filter(FilterConfig)` arms of
/// `BiquadConfig::{build, try_build}` (src/iir/config.rs:355-430); the Pid arm is `Pid` above,
/// the Raw arm is the `BiquadClamp` itself.
template <class C>
BiquadClamp<C> build_config(const idsp_ba_config &c, const Units &u, bool validate = false)
{
    typename detail::OutOf<C>::Rec r;
    check(detail::OutOf<C>::ba(&c, &u, validate, &r));
    return detail::clamp_from<C>(r);
}
template <class C>
BiquadClamp<C> build_config(const idsp_filter_config &c, const Units &u, bool validate = false)
{
    typename detail::OutOf<C>::Rec r;
    check(detail::OutOf<C>::filter(&c, &u, validate, &r));
    return detail::clamp_from<C>(r);
}

/// Same-rate linear-phase FIR `type_fir!` (src/hbf.rs:70-138): OddSymmetric / EvenSymmetric /
/// OddAntiSymmetric / EvenAntiSymmetric `<[f32; M]>` as `SplitProcess<f32, f32, [f32; N]>`.
class FirSym {
public:
    FirSym(idsp_fir_kind kind, const std::vector<float> &taps, size_t lanes, void *stream = nullptr)
        : lanes_(lanes), stream_(stream)
    {
        require(!taps.empty() && taps.size() <= IDSP_HBF_MAX_TAPS, "1..32 taps");
        cfg_.kind = kind, cfg_.m = int32_t(taps.size());
        for (size_t k = 0; k < taps.size(); k++) cfg_.taps[k] = taps[k];
        state_ = DeviceBuffer<uint32_t>(idsp_fir_sym_state_words(&cfg_) * lanes);
    }
    DeviceBuffer<uint32_t> &state() { return state_; }
    template <class Layout>
    void process_view(View<float, Layout> x, ViewMut<float, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(idsp_fir_sym_f32_process(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    idsp_fir_sym_f32 cfg_{};
    size_t lanes_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
};

// ----------------------------------------------------------------- half-band
enum class HbfTaps { Taps140 = 0 /* HBF_TAPS, hbf.rs:308-349 */, Taps98 = 1 /* HBF_TAPS_98, hbf.rs:258-292 */ };

template <bool DEC>
class HbfCascade {
public:
    /// `HBF_DEC_CASCADE` / `HBF_INT_CASCADE` restricted to a 2^stages rate change with
    /// `HbfDec2..32` / `HbfInt2..32` state for `lanes` streams (hbf.rs:363-421,454-512).
    HbfCascade(int stages, size_t lanes, HbfTaps taps = HbfTaps::Taps140, void *stream = nullptr)
        : lanes_(lanes), stream_(stream)
    {
        check(DEC ? idsp_hbf_dec_cascade(int(taps), stages, &cfg_) : idsp_hbf_int_cascade(int(taps), stages, &cfg_));
        state_ = DeviceBuffer<uint32_t>((DEC ? idsp_hbf_dec_state_words(&cfg_) : idsp_hbf_int_state_words(&cfg_)) * lanes);
    }
    size_t rate() const { return size_t(1) << cfg_.stages; }
    /// `hbf_dec_response_length` / `hbf_int_response_length` (hbf.rs:424-448,515-539)
    int response_length() const { return DEC ? idsp_hbf_dec_response_length(&cfg_) : idsp_hbf_int_response_length(&cfg_); }
    DeviceBuffer<uint32_t> &state() { return state_; }
    /// decimator: x = `[[f32; R]]` chunks, y = `[f32]`; interpolator: the reverse
    template <class Layout>
    void process_view(View<float, Layout> x, ViewMut<float, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(DEC ? idsp_hbf_dec_f32(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_)
                  : idsp_hbf_int_f32(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    idsp_hbf_cascade_f32 cfg_{};
    size_t lanes_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
};
using HbfDecCascade = HbfCascade<true>;
using HbfIntCascade = HbfCascade<false>;

/// `iir::normal::Normal<C>` (src/iir/normal.rs:28-35) x `DirectForm1`: the record is `Biquad<C>` with
/// ba = [b0, b1, b2, p.re, p.im]; `Split(normal, DirectForm1{})`-style lanes.
template <class C>
class NormalLanes {
public:
    using Sample = typename Biquad<C>::Sample;
    NormalLanes(const std::vector<Biquad<C>> &sections, size_t lanes, void *stream = nullptr)
        : lanes_(lanes), stream_(stream), state_(size_t(4) * (sizeof(Sample) / 4) * (sections.empty() ? 1 : sections.size()) * lanes)
    {
        for (const auto &c : sections) abi_.push_back(detail::to_abi(c));
    }
    /// `Normal::<f64>::from(&[[b0,b1,b2],[a0,a1,a2]])` (normal.rs:62-76) as the ba record; throws for real poles
    static Biquad<double> from_ba(const std::array<double, 6> &sos)
    {
        Biquad<double> b;
        check(idsp_normal_from_sos(sos.data(), b.ba.data()));
        return b;
    }
    DeviceBuffer<uint32_t> &state() { return state_; }
    template <class Layout>
    void process_view(View<Sample, Layout> x, ViewMut<Sample, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        if constexpr (std::is_same<Sample, int32_t>::value)
            check(idsp_normal_i32_df1(abi_.data(), abi_.size(), state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
        else if constexpr (std::is_same<Sample, float>::value)
            check(idsp_normal_f32_df1(abi_.data(), abi_.size(), state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
        else
            check(idsp_normal_f64_df1(abi_.data(), abi_.size(), state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    size_t lanes_;
    void *stream_;
    std::vector<decltype(detail::to_abi(std::declval<Biquad<C>>()))> abi_;
    DeviceBuffer<uint32_t> state_;
};

/// Serial chain of `iir::wdf::Wdf<N, M>` sections (src/iir/wdf.rs:103-171) over `lanes` lanes.
class WdfLanes {
public:
    /// `Wdf::<N, M>::quantize(&g)` (wdf.rs:126-137); throws Error(IDSP_EOUTOFRANGE) where the reference returns None
    static idsp_wdf quantize(uint32_t m, const std::vector<double> &g)
    {
        idsp_wdf w;
        check(idsp_wdf_quantize(int(g.size()), m, g.data(), &w));
        return w;
    }
    WdfLanes(const std::vector<idsp_wdf> &sections, size_t lanes, void *stream = nullptr)
        : sec_(sections), lanes_(lanes), stream_(stream), state_(idsp_wdf_state_words(sections.data(), sections.size()) * lanes)
    {
    }
    DeviceBuffer<uint32_t> &state() { return state_; }
    template <class Layout>
    void process_view(View<int32_t, Layout> x, ViewMut<int32_t, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(idsp_wdf_i32(sec_.data(), sec_.size(), state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    std::vector<idsp_wdf> sec_;
    size_t lanes_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
};

/// `Cic<T, N, M>::new(rate)` (src/cic.rs:13-47), T = int32_t or int64_t, chunked as
/// `Split::stateful(cic).decimate()` / `.interpolate()` (src/cic.rs:338-346) over `lanes` lanes.
template <class T, bool DEC>
class CicLanes {
    static_assert(std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value, "Cic<T>: T is i32 or i64");

public:
    CicLanes(int order, uint32_t rate, size_t lanes, int comb_delay = 1, void *stream = nullptr)
        : cfg_{order, comb_delay, rate}, lanes_(lanes), stream_(stream)
    {
        const size_t words = idsp_cic_state_words(&cfg_, int(sizeof(T) * 8));
        require(words > 0, "Cic: order 1..6, comb delay 1..4");
        state_ = DeviceBuffer<uint32_t>(words * lanes);
    }
    int64_t gain() const { return std::is_same<T, int64_t>::value ? idsp_cic_gain(&cfg_) : int64_t(int32_t(idsp_cic_gain(&cfg_))); }
    int gain_log2() const { return idsp_cic_gain_log2(&cfg_); }                 // cic.rs:111-113
    size_t response_length() const { return idsp_cic_response_length(&cfg_); }  // cic.rs:116-118
    size_t chunk() const { return size_t(cfg_.rate) + 1; }                      // R
    DeviceBuffer<uint32_t> &state() { return state_; }
    /// x: chunks `[T; R]` (decimator) or samples (interpolator); y the other
    template <class Layout>
    void process_view(View<T, Layout> x, ViewMut<T, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        if constexpr (std::is_same<T, int64_t>::value)
            check((DEC ? idsp_cic_dec_i64 : idsp_cic_int_i64)(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
        else
            check((DEC ? idsp_cic_dec_i32 : idsp_cic_int_i32)(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    idsp_cic cfg_;
    size_t lanes_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
};
template <class T> using CicDecimator = CicLanes<T, true>;
template <class T> using CicInterpolator = CicLanes<T, false>;

// ------------------------------------------------------------ DDS / lock-in
/// `cossin(p: i32[N]) -> i32[N, 2]` (src/py.rs:10-28)
inline void cossin(const DeviceBuffer<int32_t> &phase, DeviceBuffer<int32_t> &out, void *stream = nullptr)
{
    require(out.len() == 2 * phase.len(), "out.len() != 2 * phase.len()");
    check(idsp_cossin_i32(phase.data(), out.data(), phase.len(), stream));
}

/// `atan2(xy: i32[N, 2]) -> i32[N]` (src/py.rs:30-47); on `[re, im]` rows = `Complex<i32>::arg` (src/complex.rs:254-256)
inline void atan2(const DeviceBuffer<int32_t> &xy, DeviceBuffer<int32_t> &out, void *stream = nullptr)
{
    require(xy.len() == 2 * out.len(), "xy.len() != 2 * out.len()");
    check(idsp_atan2_i32(xy.data(), out.data(), out.len(), stream));
}

/// `Lockin<[Lowpass<N>; K]>` fed by a per-lane `Accu<Wrapping<i32>>` (src/lockin.rs:30-39).
template <int N, int K>
class Lockin {
    static_assert(N == 1 || N == 2, "Lowpass order must be 1 or 2 (src/lowpass.rs:75)");
    static_assert(K >= 1 && K <= IDSP_LOCKIN_MAX_CASCADE, "1..4 cascaded lowpasses");

public:
    Lockin(const std::array<std::array<int32_t, N>, K> &k, const std::vector<int32_t> &accu_state,
           const std::vector<int32_t> &accu_step, void *stream = nullptr)
        : lanes_(accu_step.size()), stream_(stream)
    {
        require(accu_state.size() == lanes_, "one Accu per lane");
        cfg_.order = N, cfg_.cascade = K;
        for (int c = 0; c < K; c++)
            for (int j = 0; j < N; j++) cfg_.k[c][j] = k[c][j];
        std::vector<uint32_t> st(idsp_lockin_state_words(&cfg_) * lanes_, 0u);
        for (size_t l = 0; l < lanes_; l++) st[l] = uint32_t(accu_state[l]), st[lanes_ + l] = uint32_t(accu_step[l]);
        state_ = DeviceBuffer<uint32_t>(st);
    }
    DeviceBuffer<uint32_t> &state() { return state_; }
    /// x: real samples, y: `Complex<i32>` = [re, im] per sample
    template <class Layout>
    void process_view(View<int32_t, Layout> x, ViewMut<int32_t, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(idsp_lockin_i32_process(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }
    /// Same pass with `Complex::arg()` fused in (src/complex.rs:254-256): y = one i32 phase per sample
    template <class Layout>
    void process_view_arg(View<int32_t, Layout> x, ViewMut<int32_t, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(idsp_lockin_i32_arg(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }
    /// Same pass with `Complex::norm_sqr()` fused in (src/complex.rs:214-217): y = one i64 per sample
    template <class Layout>
    void process_view_norm_sqr(View<int32_t, Layout> x, ViewMut<int64_t, Layout> y)
    {
        require(x.frames == y.frames && x.lanes == lanes_ && y.lanes == lanes_, "view shape mismatch");
        check(idsp_lockin_i32_norm_sqr(&cfg_, state_.data(), x.flat, y.flat, lanes_, x.frames, Layout::value, stream_));
    }

private:
    idsp_lockin_i32 cfg_{};
    size_t lanes_;
    void *stream_;
    DeviceBuffer<uint32_t> state_;
};

}  // namespace idsp_hip
