// dispatch_thresholds.h — every lane-count / frame-count threshold the launchers of the lane-streaming families branch on, in ONE place
// (round 5; VERDICT round 4, "upkeep").  Each was measured on MI355X boxes that differ by 3-5 %, so none of them is sharp; the file each
// comes from is named, profiles/NOTES.md has the numbers, and tests/test_gpu_dispatch_table.py pins the kernel taken at every BASELINE
// shape and on both sides of every cliff listed here.
#pragma once

#include <cstddef>

namespace idsp {
namespace thr {

// ---- FrameMajor, 4-byte inputs (launch_stream, lane_stream.h) ----------------------------------------------------------------------
// dense-sweep LDS-DMA kernel (fm_sweep.h) from this many lanes up: 8-byte outputs / 4-byte outputs (several frames per segment)
constexpr size_t kSweepMinLanes = 49152, kSweepMinLanesFps = 24576;
// ... and from this many frames up (shorter calls: the round-3 LDS-DMA kernel or the staged kernel)
constexpr size_t kSweepMinFrames = 16;
// ... and only when the launch is one sweep, or its sweeps are at least this many blocks per workgroup wide (fm_sweep.h `sweep_takes`; a processor's
// largest count is its SWEEP_MAX_LPT: 16 single biquad sections, 8 `Normal` / per-lane banks / two-section chains, 4 and 2 cascades and longer chains)
constexpr int kSweepMinLptSeveralSweeps = 8;
// rows off the 64-byte grid take the sweep kernel only above the largest single-round grid of the round-3 LDS-DMA kernel
constexpr size_t kLdsGridCap = 384;  // workgroups of 256 lanes: 98304 lanes
// ... and up to this many lanes (below: several frames per segment against the staged single-wave kernel; in between round 3's XCD-contiguous LDS-DMA kernel)
constexpr size_t kSweepOffGridSmallMax = 53248;
// "whole rounds + remainder on a second stream": remainders up to this many lanes, whole rounds of 1, 2, 4, 8 or 16 x 65536 lanes
constexpr size_t kSplitTailMax = 20480;
// staged single-wave kernel below kSweepMinLanes*: cheap processors below this lane count, heavy ones (COST > 120) inside the window
constexpr size_t kStagedMaxLanes = 49152, kStagedHeavyMinLanes = 12288, kStagedHeavyMaxLanes = 40960;
// ... lanes per wave: 64 from here (rows on / off the 64-byte grid), 32 from here, else 16
constexpr size_t kStaged64Lanes = 24576, kStaged64LanesOffGrid = 25600, kStaged32Lanes = 8192;
// compute + mover pair kernel (stream_frame_major_pair, round 6): cheap sections (COST <= kPairMaxCost) UP TO this lane count — 768 workgroups of 32 lanes and
// 32 KiB of LDS, three per CU, ahead of the sweep kernel's several-frames-per-segment form (24576 lanes 0.175 against 0.192 ms; 28672 lanes would be a second
// generation: 0.217 against 0.194) — from this many frames (four tiles; 512 frames 0.0174 against 0.0194 ms at 16384 lanes: profiles/r06_exp_fm_pair.txt)
constexpr size_t kPairMaxLanes = 24576, kPairMinFrames = 512;
constexpr int kPairMaxCost = 60;
// smallest launch (in 64-lane waves) of the round-3 LDS-DMA kernel
constexpr size_t kLdsMinWaves = 256;
// two-wave chain kernel (stream_frame_major_duo): from this many lanes, chains of 5+ sections (4 sections up to kDuo4MaxLanes)
constexpr size_t kDuoMinLanes = 40960, kDuo4MaxLanes = 98304;

// ---- LaneMajor (launch_stream) -----------------------------------------------------------------------------------------------------
// staged kernel, lanes per wave: 64 from here, 32 from here (or COST > 120), else 16
constexpr size_t kLmStaged64Lanes = 49152, kLmStaged32Lanes = 24576;

// ---- start-up stagger of the line-wise LaneMajor kernels (lockin_waves.h, dds.hip; MI355X in SPX mode only: stagger_tuned_device()) --
constexpr unsigned kStaggerMinWorkgroups = 512;                     // launches of at least two workgroups per CU
constexpr size_t kStaggerMinFrames = 2048, kStaggerMaxFrames = 8192;  // long enough to pay, short enough not to drift apart anyway
constexpr unsigned kLockinStaggerTicks = 600, kFmDiscStaggerTicks = 1200;  // 10 ns ticks between the four CU groups of an XCD

}  // namespace thr
}  // namespace idsp
