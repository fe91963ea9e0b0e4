// common.h — shared host-side plumbing of libidsp_hip.so (error reporting,
// argument checks, launch geometry).  gfx950 only; no CPU fallback exists in
// this library: every processing entry point launches a HIP kernel or fails.
#pragma once

#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "idsp_hip.h"

namespace idsp {

// thread-local last-error text (include/idsp_hip.h: idsp_last_error)
char *last_error_buf();
int fail(int code, const char *fmt, ...);

#define IDSP_HIP_TRY(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return ::idsp::fail(IDSP_EHIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// Kernel-selection switches (profiles/NOTES.md, "Kernel-selection switches") exist for A/B measurements and for tests that force a
// path a default run would not reach.  A production process must not change dispatch because of a stray
// variable, so they are honoured only when IDSP_DIAG=1 is set as well; every caller caches the answer in a
// function-local static (read once per process).
inline const char *diag_env(const char *name)
{
    static const bool on = [] {
        const char *e = getenv("IDSP_DIAG");
        return e && atoi(e) != 0;
    }();
    return on ? getenv(name) : nullptr;
}

// true when IDSP_DIAG=1: dispatch may be overridden from the environment (then nothing assumes which kernel a shape takes)
inline bool diag_on() { return diag_env("IDSP_DIAG") != nullptr; }

// A second stream per (thread, device) with the two events that fork it off the caller's stream and join it back: a launch
// can put an independent piece of a call beside the main kernel (lane_stream.h: the lanes beyond the last whole round of
// workgroups).  The device is that of the CALLER'S STREAM (hipStreamGetDevice), not the current one; a caller whose stream
// lives on another device than the current one gets NULL, and the launch takes its unsplit single-stream form.  Created on
// first use, released when the thread exits (thread_local holder); NULL if the runtime refuses anything.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
// set by an atexit handler registered at the first side_stream(): from then on the HIP runtime may already be gone, and a
// thread that ends during process teardown (a detached worker, a daemon thread joined from atexit) must not call into it
inline std::atomic<bool> &process_exiting()
{
    static std::atomic<bool> flag{false};
    return flag;
}
struct SideStreamTable {
    static constexpr int kMaxDev = 64;
    SideStream e[kMaxDev];
    ~SideStreamTable()
    {
        // The main thread's holder dies at process exit, possibly after the HIP runtime has shut down: leave its streams
        // to the process teardown.  Worker threads release theirs while the runtime is alive — i.e. before exit() began.
        if (process_exiting().load(std::memory_order_acquire)) return;
        if (getpid() == pid_t(syscall(SYS_gettid))) return;
        for (SideStream &s : e) {
            if (s.fork) (void)hipEventDestroy(s.fork);
            if (s.join) (void)hipEventDestroy(s.join);
            if (s.stream) (void)hipStreamDestroy(s.stream);
        }
    }
};
inline SideStream *side_stream(hipStream_t caller)
{
    static const bool hooked = (std::atexit([] { process_exiting().store(true, std::memory_order_release); }), true);
    (void)hooked;
    static thread_local SideStreamTable table;
    int cur = 0, dev = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (caller == nullptr)
        dev = cur;  // the NULL stream is the current device's
    else if (hipStreamGetDevice(caller, &dev) != hipSuccess)
        return nullptr;
    if (dev != cur || dev < 0 || dev >= SideStreamTable::kMaxDev) return nullptr;
    SideStream &e = table.e[dev];
    if (!e.stream) {
        hipStream_t s = nullptr;
        hipEvent_t f = nullptr, j = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        // both streams are on ONE device: no system-scope fence at the event (it costs the back-to-back launch overlap)
        constexpr unsigned kFlags = hipEventDisableTiming | hipEventDisableSystemFence;
        if (hipEventCreateWithFlags(&f, kFlags) != hipSuccess || hipEventCreateWithFlags(&j, kFlags) != hipSuccess) {
            if (f) (void)hipEventDestroy(f);
            (void)hipStreamDestroy(s);
            return nullptr;
        }
        e.stream = s, e.fork = f, e.join = j;
    }
    return &e;
}

// Name of the kernel the most recent launch on this thread dispatched to (idsp_last_kernel()).
void note_kernel(const char *kernel, const char *detail = nullptr);
// a second kernel the same call ran beside it (appended to the text; cleared by the next note_kernel())
void note_kernel_also(const char *also);
// the name the most recent note_kernel() on this thread recorded (a static string; NULL before the first launch)
const char *noted_kernel();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Checks shared by every lane-streaming entry point (the reference's
// debug_assert / const-assert preconditions, reported instead of aborting).
inline int check_stream_args(const void *cfg, size_t n, const void *state, const void *x,
                             const void *y, size_t lanes, size_t frames, int layout)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)
        return fail(IDSP_EINVAL, "layout %d is neither IDSP_FRAME_MAJOR nor IDSP_LANE_MAJOR", layout);
    if (n > IDSP_MAX_SECTIONS) return fail(IDSP_EINVAL, "n = %zu sections > IDSP_MAX_SECTIONS", n);
    if (n && !cfg) return fail(IDSP_EINVAL, "cfg is NULL");
    if (lanes && n && !state) return fail(IDSP_EINVAL, "state is NULL");
    if (lanes && frames && (!x || !y)) return fail(IDSP_EINVAL, "x or y is NULL");
    if (lanes > (size_t(1) << 31) || frames > (size_t(1) << 40))
        return fail(IDSP_EINVAL, "lanes/frames out of range");
    return IDSP_OK;
}

// Workgroup barrier that orders LDS traffic only.  HIP's __syncthreads() also
// drains vmcnt (all global loads AND stores of the wave), which serialises
// every register prefetch and every output store behind the next barrier; the
// kernels here only ever exchange data through LDS, so they wait for lgkmcnt
// alone and leave global memory operations in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Same for a single-wave workgroup: LDS operations of one wave execute in
// order, so only the compiler must be kept from moving accesses across.
__device__ __forceinline__ void lds_wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// A wave-uniform pointer pinned to SGPRs.  readfirstlane is opaque to the loop optimiser: without it the per-instruction
// addresses `uniform base + constant * tile + thread offset` are strength-reduced into one 64-bit VGPR induction pointer
// per load and per store instruction (2 x 32 pairs next to the 32 staged pieces: scratch spills).
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *q)
{
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(u)), hi = __builtin_amdgcn_readfirstlane(uint32_t(u >> 32));
    return reinterpret_cast<T *>((uint64_t(hi) << 32) | lo);
}
// 16-byte global accesses at `wave-uniform base + 32-bit thread offset` in the GLOBAL address space.  A pointer rebuilt from
// integers (uniform_ptr above) is a generic one, and hipcc then emits `flat_load / flat_store`: those count on lgkmcnt as well
// as on vmcnt — a 4-bit counter on gfx9, so a wave cannot have more than 15 of them in flight, and every `s_waitcnt lgkmcnt`
// of its LDS traffic waits for them too (round 6: both staged stream kernels ran on flat accesses).  With the address space
// named the same access is `global_load_dwordx4 v, v_off, s[base:base+1]`: vmcnt only, no 64-bit VALU address arithmetic.
template <class V, bool NT>
__device__ __forceinline__ V global_ld(const void *ubase, uint32_t voff)
{
#ifdef IDSP_EXP_FLAT  // A/B: the generic pointer of rounds 2-5 (flat_load)
    const auto *q = reinterpret_cast<const V *>(uniform_ptr(static_cast<const char *>(ubase)) + size_t(voff));
#else
    const auto *g = reinterpret_cast<const __attribute__((address_space(1))) char *>(reinterpret_cast<uintptr_t>(uniform_ptr(static_cast<const char *>(ubase))));
    const auto *q = reinterpret_cast<const __attribute__((address_space(1))) V *>(g + voff);
#endif
    if constexpr (NT)
        return __builtin_nontemporal_load(q);
    else
        return *q;
}
template <class V, bool NT>
__device__ __forceinline__ void global_st(void *ubase, uint32_t voff, V v)
{
#ifdef IDSP_EXP_FLAT
    auto *q = reinterpret_cast<V *>(uniform_ptr(static_cast<char *>(ubase)) + size_t(voff));
#else
    auto *g = reinterpret_cast<__attribute__((address_space(1))) char *>(reinterpret_cast<uintptr_t>(uniform_ptr(static_cast<char *>(ubase))));
    auto *q = reinterpret_cast<__attribute__((address_space(1))) V *>(g + voff);
#endif
    if constexpr (NT)
        __builtin_nontemporal_store(v, q);
    else
        *q = v;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the ABI lets one
// process switch devices (idsp_device_set), and the attribute is per device.
// The kernel is a template ARGUMENT (not a function parameter) so that the flag below is one per kernel: kernels of
// different processors often share one function type, and a flag per type would let the second of them skip the call.
template <auto Kernel>
inline int ensure_dyn_lds(size_t bytes)
{
    static std::atomic<uint64_t> done{0};  // one bit per device ordinal, per kernel
    int dev = 0;
    IDSP_HIP_TRY(hipGetDevice(&dev));
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return IDSP_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e != hipSuccess) return fail(IDSP_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (dev < 64) done.fetch_or(bit, std::memory_order_release);
    return IDSP_OK;
}

// The start-up stagger of the line-wise LaneMajor kernels (lockin_waves.h "lanes in phase", dds.hip) spaces the CUs of an XCD by
// ticks of the 100 MHz wall clock and relies on workgroup b running on XCD b % 8 of a 256-CU part: tuned on MI355X (gfx950) in SPX
// mode.  On any other part or partition the wait would be pure added latency, so it is armed only there (cached per device).
inline bool stagger_tuned_device()
{
    static std::atomic<int> cache[64];  // 0 unknown, 1 no, 2 yes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int c = cache[dev].load(std::memory_order_acquire);
    if (c == 0) {
        hipDeviceProp_t pr;
        c = 1;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount == 256 && strncmp(pr.gcnArchName, "gfx950", 6) == 0) c = 2;
        cache[dev].store(c, std::memory_order_release);
    }
    return c == 2;
}

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IDSP_EHIP, "kernel launch: %s", hipGetErrorString(e));
    return IDSP_OK;
}

}  // namespace idsp
