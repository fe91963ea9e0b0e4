"""Host-side mirror of the reference's processing interface for the hot path.

The reference is Rust; no Rust toolchain exists in this image, so the host
layer above the C ABI is written in Python and keeps the reference's names,
argument meaning and error behaviour so that tests read like the reference's
own:

    reference (dsp-process / idsp)                      here
    --------------------------------------------------  -------------------------------
    Biquad<Q32<F>> / Biquad<f32>   (iir/biquad.rs:96)   Biquad(ba, frac=F) / Biquad(ba)
    BiquadClamp<C,T>               (iir/biquad.rs:121)  BiquadClamp(coeff, u, min, max)
    Cascade<[Biquad; N]>           (iir/biquad.rs:324)  Cascade([...])
    DirectForm1 / DirectForm2Transposed / DirectForm1Wide / DirectForm1Dither
    Split::new(cfg, state).lanes::<N>()  (split.rs:272) Split(cfg, DirectForm1).lanes(N)
    Split::new(ByLane([c0, c1, ..]), states) (compose.rs:363) ByLane([c0, c1, ..], DirectForm1)
    Process::block(x, y) over &[[T; N]]  (process.rs:44) .block(x, y)      FrameMajor [frames, lanes]
    Inplace::inplace(xy)                 (process.rs:61) .inplace(xy)
    ViewProcess::process_view(View<LaneMajor>, ViewMut<LaneMajor>) (view.rs:245)
                                                         .process_view(View(...), ViewMut(...))
    HBF_DEC_CASCADE / HbfDec16 ...       (hbf.rs:363-421) HbfDecCascade(stages).lanes(N)
    Lockin<[Lowpass<N>; K]>, Accu        (lockin.rs, accu.rs) Lockin([...]).lanes(N, step=...)
    Lockin<C> biquad arms / external LO  (lockin.rs:16-27)    Lockin([Biquad...]), LockinLo(arms, N).process(x, lo, y)
    Split::stateful(Cic::new(rate)).decimate() (cic.rs:338) Cic(N, rate).decimate().lanes(n)
    cossin(phase)                        (cossin.rs:14)  cossin(phases)
    atan2(y, x) / Complex::arg           (atan2.rs:66)   atan2(xy)

Buffers are torch tensors on the GPU (torch is plumbing: device memory and
streams); every call goes through the C ABI of ``libidsp_hip.so`` on the
current torch stream.  Misuse that is a ``debug_assert!``/panic in the
reference raises ``ValueError`` here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import _abi
from ._lib import IdspError, call, load

__all__ = [
    "FrameMajor", "LaneMajor", "View", "ViewMut", "Biquad", "BiquadClamp", "Cascade",
    "DirectForm1", "DirectForm2Transposed", "DirectForm1Wide", "DirectForm1Dither", "DirectForm",
    "Split", "Lanes", "ByLane", "HbfDecCascade", "HbfIntCascade", "FirSym", "Cic", "Normal", "Wdf", "HBF_TAPS", "HBF_TAPS_98",
    "Lowpass", "Lockin", "LockinLo", "Accu", "Dds", "FmDisc", "cossin", "atan2", "sos", "sos_clamp_wide", "IdspError",
]

FrameMajor = _abi.FRAME_MAJOR  # dsp-process/src/view.rs:10
LaneMajor = _abi.LANE_MAJOR    # dsp-process/src/view.rs:17
HBF_TAPS = 0      # src/hbf.rs:308-349
HBF_TAPS_98 = 1   # src/hbf.rs:258-292


def _stream_ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check(t: torch.Tensor, dtype, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{what}: expected a CUDA/HIP tensor (idsp_amd has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{what}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: tensor must be contiguous")
    return t


class View:
    """`View<'_, T, Layout, L>` (dsp-process/src/view.rs:24-28): typed view of a flat
    slice; never copies or transposes."""

    mutable = False

    def __init__(self, flat: torch.Tensor, layout: int, lanes: int, frames: Optional[int] = None, width: int = 1):
        self.flat = flat.reshape(-1)
        self.layout, self.lanes, self.width = layout, lanes, width
        n = self.flat.numel()
        if frames is None:
            if lanes == 0 or n % (lanes * width):
                raise ValueError("flat.len() is not a multiple of the lane count")
            frames = n // (lanes * width)
        if n != frames * lanes * width:  # view.rs:182 assert_eq!(flat.len(), frames * L)
            raise ValueError(f"flat.len() = {n} != frames * L = {frames * lanes * width}")
        self.frames = frames


class ViewMut(View):
    """`ViewMut<'_, T, Layout, L>` (dsp-process/src/view.rs:31-36)."""

    mutable = True


# --------------------------------------------------------------------------
# configurations
# --------------------------------------------------------------------------
class Biquad:
    """`iir::Biquad<C>`: ba = [b0, b1, b2, a1, a2], a1/a2 stored as used
    (src/iir/biquad.rs:96-116).  `frac=F` selects `Q32<F>` fixed point
    (raw integer bits), `frac=None` selects `f32`."""

    def __init__(self, ba: Sequence, frac: Optional[int] = None, f64: bool = False):
        if len(ba) != 5:
            raise ValueError("ba must hold 5 coefficients")
        if frac is not None and f64:
            raise ValueError("choose Q32<F> or f64, not both")
        self.frac = frac
        self.f64 = f64  # `Biquad<f64>` instead of `Biquad<f32>`
        self.ba = [int(v) for v in ba] if frac is not None else [float(v) for v in ba]

    @property
    def is_fixed(self) -> bool:
        return self.frac is not None

    @classmethod
    def from_sos(cls, sos: Sequence[float], frac: Optional[int] = None, f32_math: bool = False, f64: bool = False) -> "Biquad":
        """`Biquad::from([[b0,b1,b2],[a0,a1,a2]])` (src/iir/biquad.rs:545-566);
        with `frac` the coefficients are quantised like `f64 -> Q32<F>`
        (dsp-fixedpoint/src/num_traits_impl.rs:32-46)."""
        load()
        if frac is not None:
            out = _abi.BiquadI32()
            call("biquad_i32_from_sos", (C.c_double * 6)(*sos), frac, C.byref(out))
            return cls(list(out.ba), frac)
        if f64:
            out = _abi.BiquadF64()
            call("biquad_f64_from_sos", (C.c_double * 6)(*sos), C.byref(out))
            return cls(list(out.ba), None, f64=True)
        out = _abi.BiquadF32()
        if f32_math:
            call("biquad_f32_from_sos", (C.c_float * 6)(*sos), C.byref(out))
        else:
            call("biquad_f32_from_sos_f64", (C.c_double * 6)(*sos), C.byref(out))
        return cls(list(out.ba), None)

    @classmethod
    def proportional(cls, k, frac: Optional[int] = None) -> "Biquad":
        """src/iir/biquad.rs:196-200"""
        return cls([k, 0, 0, 0, 0], frac)

    @classmethod
    def identity(cls, frac: Optional[int] = None) -> "Biquad":
        """`Biquad::IDENTITY` (src/iir/biquad.rs:184): proportional(C::ONE)."""
        return cls.proportional((1 << frac) if frac is not None else 1.0, frac)

    @classmethod
    def hold(cls, frac: Optional[int] = None) -> "Biquad":
        """`Biquad::HOLD` (src/iir/biquad.rs:212-214)."""
        return cls([0, 0, 0, (1 << frac) if frac is not None else 1.0, 0], frac)

    def forward_gain(self):
        """src/iir/biquad.rs:224-226"""
        return self.ba[0] + self.ba[1] + self.ba[2]


class BiquadClamp:
    """`iir::BiquadClamp<C, T>` (src/iir/biquad.rs:121-171); defaults u = 0,
    min = T::MIN, max = T::MAX (+-inf for floats, src/num.rs:33-52)."""

    def __init__(self, coeff: Biquad, u=None, min=None, max=None):
        self.coeff = coeff
        if coeff.is_fixed:
            self.u = 0 if u is None else int(u)
            self.min = -(1 << 31) if min is None else int(min)
            self.max = (1 << 31) - 1 if max is None else int(max)
        else:
            self.u = 0.0 if u is None else float(u)
            self.min = -math.inf if min is None else float(min)
            self.max = math.inf if max is None else float(max)

    @property
    def is_fixed(self) -> bool:
        return self.coeff.is_fixed

    @property
    def f64(self) -> bool:
        return self.coeff.f64


class Cascade:
    """`iir::Cascade<[Biquad<C>; N]>` (src/iir/biquad.rs:324): sections sharing delay lines."""

    def __init__(self, sections: Sequence[Biquad]):
        self.sections = list(sections)


class _StateKind:
    def __init__(self, name: str, words: int):
        self.name, self.words = name, words

    def __repr__(self):
        return self.name


DirectForm1 = _StateKind("DirectForm1", 4)                      # biquad.rs:319
DirectForm2Transposed = _StateKind("DirectForm2Transposed", 2)  # biquad.rs:407
DirectForm1Wide = _StateKind("DirectForm1Wide", 6)              # biquad.rs:445-454
DirectForm1Dither = _StateKind("DirectForm1Dither", 5)          # biquad.rs:484-491


def DirectForm(n: int) -> _StateKind:
    """`DirectForm<T, N>` (src/iir/biquad.rs:260-269), the `Cascade` state."""
    return _StateKind(f"DirectForm<{n}>", 2 + 2 * n)


class Split:
    """`Split<C, S>` (dsp-process/src/split.rs:29-34): configuration plus state
    kind; `.lanes(n)` instantiates n independent states on the GPU."""

    def __init__(self, config, state_kind: Optional[_StateKind] = None):
        self.config, self.state_kind = config, state_kind

    def lanes(self, n: int, device="cuda") -> "Lanes":
        """`Split::lanes::<N>()` (dsp-process/src/split.rs:272-277)."""
        return Lanes(self.config, self.state_kind, n, device)


def _biquad_entry(sections, kind: _StateKind):
    """Pick the C-ABI entry point + cfg array for a serial slice of sections."""
    first = sections[0]
    clamp = isinstance(first, BiquadClamp)
    if any(isinstance(s, BiquadClamp) != clamp or s.is_fixed != first.is_fixed for s in sections):
        raise ValueError("sections of one slice composition must share one type")
    fixed = first.is_fixed
    n = len(sections)
    if fixed:
        table = {"DirectForm1": "biquad_i32_df1", "DirectForm1Dither": "biquad_i32_dither",
                 "DirectForm1Wide": "biquad_i32_wide"}
        if kind.name not in table:
            raise ValueError(f"no SplitProcess impl for Biquad<Q32<F>> on {kind}")
        name = table[kind.name] + ("_clamp" if clamp else "")
        if clamp:
            arr = (_abi.BiquadClampI32 * n)()
            for a, s in zip(arr, sections):
                a.ba[:] = s.coeff.ba
                a.frac, a.u, a.min, a.max = s.coeff.frac, s.u, s.min, s.max
        else:
            arr = (_abi.BiquadI32 * n)()
            for a, s in zip(arr, sections):
                a.ba[:] = s.ba
                a.frac = s.frac
        return name, arr, torch.int32
    table = {"DirectForm1": "biquad_f32_df1", "DirectForm2Transposed": "biquad_f32_df2t"}
    if kind.name not in table:
        raise ValueError(f"no SplitProcess impl for Biquad<f32> on {kind}")
    name = table[kind.name] + ("_clamp" if clamp else "")
    if first.f64:
        if any(not s.f64 for s in sections):
            raise ValueError("sections of one slice composition must share one type")
        name = name.replace("f32", "f64")
        if clamp:
            arr = (_abi.BiquadClampF64 * n)()
            for a, s in zip(arr, sections):
                a.ba[:] = s.coeff.ba
                a.u, a.min, a.max = s.u, s.min, s.max
        else:
            arr = (_abi.BiquadF64 * n)()
            for a, s in zip(arr, sections):
                a.ba[:] = s.ba
        return name, arr, torch.float64
    if clamp:
        arr = (_abi.BiquadClampF32 * n)()
        for a, s in zip(arr, sections):
            a.ba[:] = s.coeff.ba
            a.u, a.min, a.max = s.u, s.min, s.max
    else:
        arr = (_abi.BiquadF32 * n)()
        for a, s in zip(arr, sections):
            a.ba[:] = s.ba
    return name, arr, torch.float32


class _LaneOp:
    """Common machinery: per-lane state on the device + both layouts."""

    dtype_in = torch.int32
    dtype_out = torch.int32
    in_width = 1   # elements per input sample  (R for the decimator chunk type)
    out_width = 1  # elements per output sample (2 for Complex, R for the interpolator)

    def __init__(self, n_lanes: int, words: int, device):
        load()
        self.n_lanes = int(n_lanes)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("idsp_amd runs on the GPU only")
        # zero words == the reference's `Default::default()` state
        self.state = torch.zeros((words, self.n_lanes), dtype=torch.int32, device=self.device)

    # -- subclasses implement the actual ABI call
    def _run(self, x, y, frames: int, layout: int):  # pragma: no cover - abstract
        raise NotImplementedError

    def _frames_fm(self, t: torch.Tensor, width: int, what: str) -> int:
        expect = self.n_lanes * width
        if t.dim() < 2 or t.numel() % max(expect, 1) or (self.n_lanes and t.shape[1] != self.n_lanes):
            raise ValueError(f"{what}: expected [frames, {self.n_lanes}{', ' + str(width) if width > 1 else ''}]")
        return t.numel() // expect if expect else 0

    def block(self, x: torch.Tensor, y: torch.Tensor):
        """`Process::block(&mut self, x: &[[X; N]], y: &mut [[Y; N]])`
        (dsp-process/src/process.rs:44-49 on `Lanes`, compose.rs:468-476):
        FrameMajor, x[frame, lane(, k)]."""
        _check(x, self.dtype_in, "x")
        _check(y, self.dtype_out, "y")
        frames = self._frames_fm(x, self.in_width, "x")
        if self._frames_fm(y, self.out_width, "y") != frames:
            raise ValueError("x.len() != y.len()")  # process.rs:45 debug_assert_eq!
        self._run(x, y, frames, FrameMajor)
        return y

    def inplace(self, xy: torch.Tensor):
        """`Inplace::inplace(&mut self, xy: &mut [[X; N]])` (process.rs:61-65)."""
        if self.in_width != self.out_width or self.dtype_in != self.dtype_out:
            raise ValueError("inplace needs identical input and output types")
        return self.block(xy, xy)

    def process_view(self, x: View, y: ViewMut):
        """`ViewProcess::process_view` (dsp-process/src/view.rs:245-248); LaneMajor
        views take the `Lanes` lane-slice path (compose.rs:478-494), FrameMajor
        views fall back to `block` (view.rs:268-283)."""
        if not y.mutable:
            raise ValueError("y must be a ViewMut")
        if x.layout != y.layout or x.lanes != self.n_lanes or y.lanes != self.n_lanes:
            raise ValueError("view layout / lane count mismatch")
        if x.width != self.in_width or y.width != self.out_width:
            raise ValueError("view element width mismatch")
        if x.frames != y.frames:
            raise ValueError("x.frames() != y.frames()")  # compose.rs:488
        _check(x.flat, self.dtype_in, "x")
        _check(y.flat, self.dtype_out, "y")
        self._run(x.flat, y.flat, x.frames, x.layout)

    def inplace_view(self, xy: ViewMut):
        """`ViewInplace::inplace_view` (dsp-process/src/view.rs:251-254)."""
        self.process_view(xy, xy)

    def reset(self):
        self.state.zero_()


class Lanes(_LaneOp):
    """`Split<Lanes<C>, [S; N]>` for the biquad family: one shared configuration,
    N states (dsp-process/src/compose.rs:449-513).  `config` is a Biquad, a
    BiquadClamp, a list of either (`[C] x [S]` serial slice composition,
    compose.rs:43-77) or a Cascade."""

    def __init__(self, config, state_kind: Optional[_StateKind], n_lanes: int, device="cuda"):
        if isinstance(config, Cascade):
            secs = config.sections
            if not secs:
                raise ValueError("Cascade needs at least one section")
            kind = DirectForm(len(secs))
            if state_kind is not None and state_kind.words != kind.words:
                raise ValueError("Cascade<[Biquad; N]> needs DirectForm<T, N>")
            fixed = secs[0].is_fixed
            self._name = "cascade_i32_df1" if fixed else ("cascade_f64_df1" if secs[0].f64 else "cascade_f32_df1")
            if fixed:
                arr = (_abi.BiquadI32 * len(secs))()
                for a, s in zip(arr, secs):
                    a.ba[:] = s.ba
                    a.frac = s.frac
            elif secs[0].f64:
                arr = (_abi.BiquadF64 * len(secs))()
                for a, s in zip(arr, secs):
                    a.ba[:] = s.ba
            else:
                arr = (_abi.BiquadF32 * len(secs))()
                for a, s in zip(arr, secs):
                    a.ba[:] = s.ba
            self._cfg, dt = arr, (torch.int32 if fixed else (torch.float64 if secs[0].f64 else torch.float32))
            self._n = len(secs)
            words = kind.words * (2 if dt == torch.float64 else 1)
        else:
            secs = list(config) if isinstance(config, (list, tuple)) else [config]
            if state_kind is None:
                raise ValueError("a state kind (DirectForm1, ...) is required")
            self._n = len(secs)
            if secs:
                self._name, self._cfg, dt = _biquad_entry(secs, state_kind)
            else:  # empty slice: identity (compose.rs:63-65); dtype fixed by the caller
                self._name, self._cfg, dt = "biquad_i32_df1", None, torch.int32
            words = state_kind.words * max(self._n, 1) * (2 if dt == torch.float64 else 1)
        self.dtype_in = self.dtype_out = dt
        super().__init__(n_lanes, words, device)

    def _run(self, x, y, frames, layout):
        call(self._name, C.cast(self._cfg, C.c_void_p) if self._cfg is not None else None, self._n,
             C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
             self.n_lanes, frames, layout, _stream_ptr(x))


class ByLane(_LaneOp):
    """`Split<ByLane<[C; N]>, [S; N]>` (dsp-process/src/compose.rs:363-390): lane i is
    filtered by configuration i with state i.  `configs[i]` is a Biquad, a BiquadClamp or
    a list of either (serial slice, compose.rs:43-77); all lanes share one type, section
    count and (for `Q32<F>`) one F.  The coefficients live on the GPU as lane-contiguous
    planes (`self.coef`, [sections, 5 or 8, lanes]) and may be rewritten between calls
    with `set_lane`."""

    def __init__(self, configs: Sequence, state_kind: _StateKind, device="cuda"):
        rows = [list(c) if isinstance(c, (list, tuple)) else [c] for c in configs]
        if not rows or not rows[0]:
            raise ValueError("ByLane needs at least one lane and one section")
        first = rows[0][0]
        self._clamp = isinstance(first, BiquadClamp)
        self._n = len(rows[0])
        self._frac = (first.coeff.frac if self._clamp else first.frac)
        self._f64 = first.f64
        name, _, dt = _biquad_entry(rows[0], state_kind)  # validates the (config, state) pairing
        self._name = name + "_bylane"
        self.dtype_in = self.dtype_out = dt
        super().__init__(len(rows), state_kind.words * self._n * (2 if dt == torch.float64 else 1), device)
        self.coef = torch.zeros((self._n, 8 if self._clamp else 5, self.n_lanes), dtype=dt, device="cpu")
        for i, r in enumerate(rows):
            self._fill(i, r)
        self.coef = self.coef.to(self.device)

    def _fill(self, lane: int, row):
        if len(row) != self._n:
            raise ValueError("every lane needs the same number of sections")
        for k, s in enumerate(row):
            c = s.coeff if self._clamp else s
            if isinstance(s, BiquadClamp) != self._clamp or c.frac != self._frac or c.f64 != self._f64:
                raise ValueError("all lanes of a ByLane must share one configuration type")
            vals = list(c.ba) + ([s.u, s.min, s.max] if self._clamp else [])
            self.coef[k, :, lane] = torch.tensor(vals, dtype=self.coef.dtype)

    def set_lane(self, lane: int, config):
        """Replace the configuration of one lane (its state is kept, like assigning `by_lane.0[i]`)."""
        row = list(config) if isinstance(config, (list, tuple)) else [config]
        host = torch.zeros((self._n, self.coef.shape[1], 1), dtype=self.coef.dtype)
        dev, self.coef = self.coef, host
        try:
            self._fill(0, row)
        finally:
            host, self.coef = self.coef, dev
        self.coef[:, :, lane] = host[:, :, 0].to(self.device)

    def _run(self, x, y, frames, layout):
        args = (C.c_void_p(self.coef.data_ptr()),) + ((self._frac,) if self._frac is not None else ()) + (
            self._n, C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
            self.n_lanes, frames, layout, _stream_ptr(x))
        call(self._name, *args)


# --------------------------------------------------------------------------
# half-band cascades
# --------------------------------------------------------------------------
class _Hbf(_LaneOp):
    dtype_in = dtype_out = torch.float32
    _dec = True

    def __init__(self, stages: int = None, tap_set: int = HBF_TAPS, taps: Optional[Sequence[Sequence[float]]] = None):
        load()
        self.cfg = _abi.HbfCascadeF32()
        if taps is not None:  # explicit `EvenSymmetric<[f32; M]>` stages in processing order
            if not 1 <= len(taps) <= _abi.HBF_MAX_STAGES:
                raise ValueError("1..5 stages")
            self.cfg.stages = len(taps)
            for s, t in enumerate(taps):
                if not 1 <= len(t) <= _abi.HBF_MAX_TAPS:
                    raise ValueError("1..32 taps per stage")
                self.cfg.m[s] = len(t)
                for k, v in enumerate(t):
                    self.cfg.taps[s][k] = v
        else:
            call("hbf_dec_cascade" if self._dec else "hbf_int_cascade", tap_set, stages, C.byref(self.cfg))
        self.stages = self.cfg.stages
        self.rate = 1 << self.stages

    def response_length(self) -> int:
        """`hbf_dec_response_length` / `hbf_int_response_length` (src/hbf.rs:424-448,515-539)."""
        return call("hbf_dec_response_length" if self._dec else "hbf_int_response_length", C.byref(self.cfg))

    def lanes(self, n: int, device="cuda"):
        words = call("hbf_dec_state_words" if self._dec else "hbf_int_state_words", C.byref(self.cfg))
        _LaneOp.__init__(self, n, words, device)
        if self._dec:
            self.in_width, self.out_width = self.rate, 1
        else:
            self.in_width, self.out_width = 1, self.rate
        return self

    def _run(self, x, y, frames, layout):
        call("hbf_dec_f32" if self._dec else "hbf_int_f32", C.byref(self.cfg), C.c_void_p(self.state.data_ptr()),
             C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class HbfDecCascade(_Hbf):
    """`HBF_DEC_CASCADE` restricted to a 2^stages decimation with `HbfDec2..32`
    state (src/hbf.rs:363-421).  Input element `[f32; R]`, output `f32`."""

    _dec = True


class HbfIntCascade(_Hbf):
    """`HBF_INT_CASCADE` / `HbfInt2..32` (src/hbf.rs:454-512).  Input `f32`, output `[f32; R]`."""

    _dec = False


class FirSym(_LaneOp):
    """Same-rate linear-phase FIR `OddSymmetric / EvenSymmetric / OddAntiSymmetric /
    EvenAntiSymmetric<[f32; M]>` as `SplitProcess<f32, f32, [f32; N]>` (src/hbf.rs:70-138)."""

    dtype_in = dtype_out = torch.float32
    KINDS = {"OddSymmetric": 0, "EvenSymmetric": 1, "OddAntiSymmetric": 2, "EvenAntiSymmetric": 3}

    def __init__(self, kind: str, taps: Sequence[float]):
        load()
        if kind not in self.KINDS or not 1 <= len(taps) <= _abi.HBF_MAX_TAPS:
            raise ValueError("kind must name one of the four type_fir! types, 1..32 taps")
        self.cfg = _abi.FirSymF32()
        self.cfg.kind, self.cfg.m = self.KINDS[kind], len(taps)
        for k, v in enumerate(taps):
            self.cfg.taps[k] = v

    def lanes(self, n: int, device="cuda") -> "FirSym":
        _LaneOp.__init__(self, n, call("fir_sym_state_words", C.byref(self.cfg)), device)
        return self

    def _run(self, x, y, frames, layout):
        call("fir_sym_f32_process", C.byref(self.cfg), C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class Normal:
    """`iir::normal::Normal<C>` (src/iir/normal.rs:28-35): feed-forward b[3] and the pole p = re + j im.
    `frac=F`: `Normal<Q32<F>>` on i32 samples (raw bits), else f32, or f64 with `f64=True`."""

    def __init__(self, b: Sequence, p: Sequence, frac: Optional[int] = None, f64: bool = False):
        if len(b) != 3 or len(p) != 2:
            raise ValueError("Normal: b has 3 entries, p = (re, im)")
        conv = int if frac is not None else float
        self.ba = [conv(v) for v in list(b) + list(p)]
        self.frac, self.f64 = frac, f64

    @classmethod
    def from_ba(cls, ba: Sequence[Sequence[float]]) -> "Normal":
        """`Normal::<f64>::from(&[[b0,b1,b2],[a0,a1,a2]])` (normal.rs:62-76); real poles raise like the assert."""
        load()
        out = (C.c_double * 5)()
        call("normal_from_sos", (C.c_double * 6)(*(list(ba[0]) + list(ba[1]))), out)
        return cls(list(out)[:3], list(out)[3:], f64=True)

    def lanes(self, n: int, device="cuda") -> "_NormalLanes":
        """`Split::new(normal, DirectForm1::default()).lanes::<N>()`"""
        return _NormalLanes([self], n, device)


class _NormalLanes(_LaneOp):
    def __init__(self, sections: Sequence[Normal], n: int, device):
        first = sections[0]
        self._n = len(sections)
        if first.frac is not None:
            self._cfg = (_abi.BiquadI32 * self._n)()
            for a, s in zip(self._cfg, sections):
                a.ba[:] = s.ba
                a.frac = s.frac
            self._name, dt = "normal_i32_df1", torch.int32
        elif first.f64:
            self._cfg = (_abi.BiquadF64 * self._n)()
            for a, s in zip(self._cfg, sections):
                a.ba[:] = s.ba
            self._name, dt = "normal_f64_df1", torch.float64
        else:
            self._cfg = (_abi.BiquadF32 * self._n)()
            for a, s in zip(self._cfg, sections):
                a.ba[:] = s.ba
            self._name, dt = "normal_f32_df1", torch.float32
        self.dtype_in = self.dtype_out = dt
        super().__init__(n, 4 * self._n * (2 if dt == torch.float64 else 1), device)

    def _run(self, x, y, frames, layout):
        call(self._name, C.cast(self._cfg, C.c_void_p), self._n, C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class Wdf:
    """`iir::wdf::Wdf<N, M>` (src/iir/wdf.rs:103-137): N two-port adaptors typed by the nibbles of M.
    `Wdf.quantize(m, g)` is `Wdf::<N, M>::quantize(&g)` (None when a pole does not fit its adaptor);
    a list of sections passed to `Wdf.chain([...]).lanes(n)` runs them in series."""

    def __init__(self, cfg: _abi.Wdf):
        self.cfg = cfg

    @classmethod
    def quantize(cls, m: int, g: Sequence[float]) -> Optional["Wdf"]:
        load()
        out = _abi.Wdf()
        rc = load()[0]["wdf_quantize"](len(g), m, (C.c_double * len(g))(*g), C.byref(out))
        if rc == _abi.IDSP_EOUTOFRANGE:
            return None
        if rc < 0:
            raise ValueError("Wdf: order 1..8")
        return cls(out)

    @classmethod
    def default(cls, n: int, m: int) -> "Wdf":
        """`Wdf::<N, M>::default()` (wdf.rs:108-114): zero coefficients"""
        c = _abi.Wdf()
        c.n, c.m = n, m
        return cls(c)

    @staticmethod
    def chain(sections: Sequence["Wdf"]) -> "_WdfChain":
        return _WdfChain(list(sections))

    def lanes(self, n: int, device="cuda") -> "_WdfChain":
        return _WdfChain([self]).lanes(n, device)


class _WdfChain(_LaneOp):
    dtype_in = dtype_out = torch.int32

    def __init__(self, sections: Sequence[Wdf]):
        load()
        self._n = len(sections)
        self._cfg = (_abi.Wdf * max(self._n, 1))()
        for d, s in zip(self._cfg, sections):
            d.n, d.m = s.cfg.n, s.cfg.m
            d.a[:] = list(s.cfg.a)

    def lanes(self, n: int, device="cuda") -> "_WdfChain":
        words = call("wdf_state_words", C.cast(self._cfg, C.c_void_p), self._n)
        _LaneOp.__init__(self, n, max(words, 1), device)
        return self

    def _run(self, x, y, frames, layout):
        call("wdf_i32", C.cast(self._cfg, C.c_void_p), self._n, C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class Cic:
    """`Cic<T, N, M>::new(rate)` (src/cic.rs:13-47) with T = i64 (default) or i32.  `.decimate()` /
    `.interpolate()` give the chunked processors of src/cic.rs:338-346, `.lanes(n)` puts n of them on the GPU."""

    def __init__(self, order: int, rate: int, comb_delay: int = 1, dtype=torch.int64):
        load()
        if dtype not in (torch.int32, torch.int64):
            raise ValueError("Cic<T>: T is i32 or i64")
        self.cfg = _abi.Cic(int(order), int(comb_delay), int(rate))
        self.dtype = dtype
        if call("cic_state_words", C.byref(self.cfg), 64) == 0:
            raise ValueError("Cic: order 1..6, comb delay 1..4 (src/cic.rs:36: must be non-zero)")

    def order(self) -> int:  # cic.rs:57-59
        return self.cfg.order

    def comb_delay(self) -> int:  # cic.rs:62-64
        return self.cfg.comb_delay

    def rate(self) -> int:  # cic.rs:69-71
        return self.cfg.rate

    def gain(self) -> int:
        """`Cic::gain()` (cic.rs:103-105) in T (wrapping)"""
        g = call("cic_gain", C.byref(self.cfg))
        return g if self.dtype == torch.int64 else ((g + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)

    def gain_log2(self) -> int:  # cic.rs:111-113
        return call("cic_gain_log2", C.byref(self.cfg))

    def response_length(self) -> int:  # cic.rs:116-118
        return call("cic_response_length", C.byref(self.cfg))

    def decimate(self) -> "_CicLanes":
        """`Split::stateful(cic).decimate()`: `Process<[T; R], T>`, R = rate + 1"""
        return _CicLanes(self, True)

    def interpolate(self) -> "_CicLanes":
        """`Split::stateful(cic).interpolate()`: `Process<T, [T; R]>`"""
        return _CicLanes(self, False)


class _CicLanes(_LaneOp):
    def __init__(self, cic: Cic, dec: bool):
        self.cic, self._dec = cic, dec
        self.dtype_in = self.dtype_out = cic.dtype
        r = cic.cfg.rate + 1
        self.in_width, self.out_width = (r, 1) if dec else (1, r)
        self._name = ("cic_dec_" if dec else "cic_int_") + ("i64" if cic.dtype == torch.int64 else "i32")

    def lanes(self, n: int, device="cuda") -> "_CicLanes":
        words = call("cic_state_words", C.byref(self.cic.cfg), 64 if self.cic.dtype == torch.int64 else 32)
        _LaneOp.__init__(self, n, words, device)
        return self

    def inplace(self, xy):
        raise ValueError("a rate changer has no in-place form")

    def _run(self, x, y, frames, layout):
        call(self._name, C.byref(self.cic.cfg), C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


# --------------------------------------------------------------------------
# DDS / lock-in
# --------------------------------------------------------------------------
class Lowpass:
    """`Lowpass<N>(pub [i32; N])` (src/lowpass.rs:13), N in {1, 2}."""

    def __init__(self, k: Sequence[int]):
        if len(k) not in (1, 2):
            raise NotImplementedError("Lowpass order must be 1 or 2")  # lowpass.rs:75 unimplemented!()
        self.k = [int(v) for v in k]


def _lockin_cfg(lowpasses: Sequence[Lowpass]) -> _abi.LockinI32:
    if not 1 <= len(lowpasses) <= _abi.LOCKIN_MAX_CASCADE:
        raise ValueError("1..4 cascaded lowpasses")
    order = len(lowpasses[0].k)
    if any(len(lp.k) != order for lp in lowpasses):
        raise ValueError("[Lowpass<N>; K]: all elements share N")
    cfg = _abi.LockinI32()
    cfg.order, cfg.cascade = order, len(lowpasses)
    for c, lp in enumerate(lowpasses):
        for j, v in enumerate(lp.k):
            cfg.k[c][j] = v
    return cfg


class Accu:
    """`Accu<Wrapping<i32>>` (src/accu.rs:16-41)."""

    def __init__(self, state: int, step: int):
        self.state, self.step = state, step


def _to_i32_tensor(v, n, device):
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=torch.int32).reshape(n)
    t = torch.tensor(v, dtype=torch.int64).reshape(-1)
    t = ((t + (1 << 31)) % (1 << 32)) - (1 << 31)
    return t.to(torch.int32).expand(n).contiguous().to(device)


class LowpassLanes(_LaneOp):
    """`Split<Lanes<[Lowpass<N>; K]>, [[LowpassState<N>; K]; lanes]>`."""

    def __init__(self, lowpasses: Sequence[Lowpass], n_lanes: int, device="cuda"):
        self.cfg = _lockin_cfg(lowpasses)
        super().__init__(n_lanes, 2 * self.cfg.order * self.cfg.cascade, device)

    def _run(self, x, y, frames, layout):
        call("lowpass_i32", C.byref(self.cfg), C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class Lockin(_LaneOp):
    """`Lockin<[Lowpass<N>; K]>` fed by a per-lane `Accu` phase accumulator:
    per sample `Lockin::process(state, (x, Wrapping(accu.next())))`
    (src/lockin.rs:30-39).  Output element `Complex<i32>` = [re, im] (`output="iq"`), or with the
    polar read-out fused into the same pass: `output="arg"` -> `Complex::arg()` as i32
    (src/complex.rs:254-256), `output="norm_sqr"` -> `Complex::norm_sqr()` as i64 (src/complex.rs:214-217)."""

    _OUTPUTS = {"iq": ("lockin_i32_process", 2, torch.int32), "arg": ("lockin_i32_arg", 1, torch.int32),
                "norm_sqr": ("lockin_i32_norm_sqr", 1, torch.int64)}

    def __init__(self, arms: Sequence, output: str = "iq"):
        """`arms`: `[Lowpass<N>; K]`, or — `Lockin<C>` takes any arm filter (src/lockin.rs:16-27) — 1..4 fixed-point
        `Biquad`s (`[Biquad<Q32<F>>; n]` x `[DirectForm1<i32>; n]`, the same sections on I and Q; `output="iq"` only)."""
        if output not in self._OUTPUTS:
            raise ValueError(f"output must be one of {sorted(self._OUTPUTS)}")
        self.biquads = None
        if arms and isinstance(arms[0], Biquad):
            if output != "iq" or not all(b.is_fixed for b in arms) or not 1 <= len(arms) <= _abi.LOCKIN_MAX_SECTIONS:
                raise ValueError("biquad arms: 1..4 Biquad<Q32<F>> sections, output 'iq'")
            self.biquads = (_abi.BiquadI32 * len(arms))()
            for rec, b in zip(self.biquads, arms):
                rec.ba[:] = b.ba
                rec.frac = b.frac
            self._entry, self.out_width, self.dtype_out = "lockin_i32_biquad_process", 2, torch.int32
            return
        self.cfg = _lockin_cfg(arms)
        self._entry, self.out_width, self.dtype_out = self._OUTPUTS[output]

    def lanes(self, n: int, step, state=0, device="cuda") -> "Lockin":
        words = (call("lockin_biquad_state_words", len(self.biquads), 1) if self.biquads is not None
                 else call("lockin_state_words", C.byref(self.cfg)))
        _LaneOp.__init__(self, n, words, device)
        self.state[0] = _to_i32_tensor(state, n, self.device)
        self.state[1] = _to_i32_tensor(step, n, self.device)
        return self

    def _run(self, x, y, frames, layout):
        if self.biquads is not None:
            call(self._entry, C.cast(self.biquads, C.c_void_p), len(self.biquads), C.c_void_p(self.state.data_ptr()),
                 C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))
            return
        call(self._entry, C.byref(self.cfg), C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


class LockinLo:
    """`Lockin<C>` on `(x, Complex<U>)` (src/lockin.rs:17-27): the LO is an input, not a phase.  Arms: `[Lowpass<N>; K]`,
    1..4 `Biquad<Q32<F>>` (x: i32, lo: `Complex<Q32<32>>` bits) or 1..4 `Biquad<f32>` (x, lo: f32 — with lo = (cos, -sin) the
    `mix * lowpass.lanes()` graph of examples/ddc_lockin.rs:35-42).  State: `[S; 2]` per lane, zero = `Default`."""

    def __init__(self, arms: Sequence, n_lanes: int, device="cuda"):
        load()
        self.n_lanes, self.device = int(n_lanes), torch.device(device)
        if arms and isinstance(arms[0], Biquad):
            if not 1 <= len(arms) <= _abi.LOCKIN_MAX_SECTIONS or any(b.f64 or b.is_fixed != arms[0].is_fixed for b in arms):
                raise ValueError("1..4 Biquad sections, all Q32<F> or all f32")
            fixed = arms[0].is_fixed
            self.cfg = ((_abi.BiquadI32 if fixed else _abi.BiquadF32) * len(arms))()
            for rec, b in zip(self.cfg, arms):
                rec.ba[:] = b.ba
                if fixed:
                    rec.frac = b.frac
            self.n, self.dtype = len(arms), torch.int32 if fixed else torch.float32
            self._entry = "lockin_i32_biquad_lo_process" if fixed else "lockin_f32_biquad_lo_process"
            words = call("lockin_biquad_state_words", self.n, 0)
        else:
            self.cfg, self.n, self.dtype, self._entry = _lockin_cfg(arms), None, torch.int32, "lockin_i32_lo_process"
            words = call("lockin_state_words", C.byref(self.cfg)) - 2
        self.state = torch.zeros((words, self.n_lanes), dtype=torch.int32, device=self.device)

    def process(self, x: torch.Tensor, lo: torch.Tensor, y: torch.Tensor, layout: int = FrameMajor):
        """x[frames, lanes] (FrameMajor) or [lanes, frames]; lo and y carry a trailing [re, im] pair per sample."""
        for t, what in ((x, "x"), (lo, "lo"), (y, "y")):
            _check(t, self.dtype, what)
        if x.numel() % max(self.n_lanes, 1) or lo.numel() != 2 * x.numel() or y.numel() != 2 * x.numel():
            raise ValueError("x.len() != lo.len() != y.len()")
        frames = x.numel() // self.n_lanes if self.n_lanes else 0
        head = (C.byref(self.cfg),) if self.n is None else (C.cast(self.cfg, C.c_void_p), self.n)
        call(self._entry, *head, C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(lo.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))
        return y


class Dds:
    """Per-lane `Accu<Wrapping<i32>>` -> `Complex::<i32>::from_angle`
    (src/accu.rs:34-41, src/complex.rs:237-240)."""

    def __init__(self, n_lanes: int, step, state=0, device="cuda"):
        load()
        self.n_lanes, self.device = n_lanes, torch.device(device)
        self.state = torch.zeros((2, n_lanes), dtype=torch.int32, device=self.device)
        self.state[0] = _to_i32_tensor(state, n_lanes, self.device)
        self.state[1] = _to_i32_tensor(step, n_lanes, self.device)

    def generate(self, out: torch.Tensor, frames: int, layout: int = FrameMajor):
        _check(out, torch.int32, "out")
        if out.numel() != frames * self.n_lanes * 2:
            raise ValueError("out.len() != frames * lanes * 2")
        call("dds_i32", C.c_void_p(self.state.data_ptr()), C.c_void_p(out.data_ptr()), self.n_lanes, frames, layout,
             _stream_ptr(out))
        return out


class FmDisc(_LaneOp):
    """The receiver core of examples/fm_disc.rs:25-50: `(Split::new(FmDiscriminator { carrier }, None) *
    Split::new(deemph, DirectForm1::default())).minor()` with `deemph: Biquad<Q32<F>>`.  Input
    `Complex<Q32<32>>` bits as [..., 2] i32, output i32 phase increments (2^32 = one turn per sample)."""

    dtype_in = dtype_out = torch.int32
    in_width = 2

    def __init__(self, carrier: int, deemph: Biquad):
        load()
        if not deemph.is_fixed:
            raise ValueError("deemph must be a Biquad<Q32<F>>")
        self.cfg = _abi.FmDisc()
        self.cfg.carrier = ((int(carrier) + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
        self.cfg.deemph.ba[:] = deemph.ba
        self.cfg.deemph.frac = deemph.frac

    def lanes(self, n: int, device="cuda") -> "FmDisc":
        _LaneOp.__init__(self, n, _abi.FM_DISC_STATE_WORDS, device)
        return self

    def inplace(self, xy):
        raise ValueError("Complex in, real out: no in-place form")

    def _run(self, x, y, frames, layout):
        call("fm_disc_i32", C.byref(self.cfg), C.c_void_p(self.state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), self.n_lanes, frames, layout, _stream_ptr(x))


def cossin(p: torch.Tensor) -> torch.Tensor:
    """`cossin(p: i32[N]) -> i32[N, 2]` — the pyo3 function of src/py.rs:10-28."""
    _check(p, torch.int32, "p")
    out = torch.empty((p.numel(), 2), dtype=torch.int32, device=p.device)
    call("cossin_i32", C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()), p.numel(), _stream_ptr(p))
    return out


def atan2(xy: torch.Tensor) -> torch.Tensor:
    """`atan2(xy: i32[N, 2]) -> i32[N]` — the pyo3 function of src/py.rs:30-47; rows
    are `[x, y]`, the result is the angle in turns (2^31 = pi).  On a `[re, im]` tensor
    this is `Complex<i32>::arg` (src/complex.rs:254-256)."""
    _check(xy, torch.int32, "xy")
    if xy.dim() != 2 or xy.shape[1] != 2:
        raise ValueError("xy must have shape [N, 2]")
    out = torch.empty(xy.shape[0], dtype=torch.int32, device=xy.device)
    call("atan2_i32", C.c_void_p(xy.data_ptr()), C.c_void_p(out.data_ptr()), out.numel(), _stream_ptr(xy))
    return out


def sos(sos_rows: Sequence[Sequence[float]], xy: torch.Tensor, lanes: int = 1, layout: int = LaneMajor):
    """`sos(sos: f64[K,6], xy: i32[N])` of src/py.rs:49-73: quantise K sections to
    Q29 and filter `xy` in place through `[Biquad<Q32<29>>] x [DirectForm1]`
    with fresh state.  `lanes > 1` treats `xy` as that many independent
    streams in the given layout (the batched drop-in)."""
    secs = [Biquad.from_sos(row, frac=29) for row in sos_rows]
    p = Lanes(secs, DirectForm1, lanes, xy.device)
    p.inplace_view(ViewMut(xy, layout, lanes))
    return xy


def sos_clamp_wide(sos_rows: Sequence[Sequence[float]], xy: torch.Tensor, lanes: int = 1, layout: int = LaneMajor):
    """`sos_clamp_wide(sos: f64[K,9], xy)` of src/py.rs:76-108: rows are
    [b0,b1,b2,a0,a1,a2,u,min,max]; u/min/max are `round()`ed then cast `as i32`."""

    def rnd(v: float) -> int:  # f64::round (half away from zero) then saturating `as i32`
        if math.isnan(v):
            return 0
        r = math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1) if math.isfinite(v) else v
        return int(max(-(1 << 31), min((1 << 31) - 1, r)))

    secs = [BiquadClamp(Biquad.from_sos(row[:6], frac=29), rnd(row[6]), rnd(row[7]), rnd(row[8])) for row in sos_rows]
    p = Lanes(secs, DirectForm1Wide, lanes, xy.device)
    p.inplace_view(ViewMut(xy, layout, lanes))
    return xy
