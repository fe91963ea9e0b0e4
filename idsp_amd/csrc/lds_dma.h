// lds_dma.h — gfx950 LDS-DMA requests (`global_load_lds_dwordx4`: 64 lanes x 16 bytes from global memory straight into
// 1 KiB of LDS, no VGPR in between) and the hand-counted wait that goes with them.  hipcc neither counts an asm load nor
// waits for it (the kernels' own `s_waitcnt vmcnt(N)` do), and M0 — the LDS destination base — is compiler-reserved, so
// every request saves, sets and restores it inside ONE asm statement.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace idsp {

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    // LDS destination = wave-uniform byte address (M0) + lane * 16; `nt`: streamed once, do not
    // keep it in L2/MALL (+5 % with nontemporal loads and stores, tools/exp_lds.hip)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))  // uniform by construction; pin it to an SGPR
                 : "memory");
}
// Same with the address split into a wave-uniform base (SGPR pair) and a 32-bit thread offset: no 64-bit VALU add per request
__device__ __forceinline__ void glds16_s(const void *sbase, uint32_t voff, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
// plain (cacheable) form: rows off the 64-byte grid, where neighbouring workgroups share lines (XCDC)
__device__ __forceinline__ void glds16_plain(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
// plain form with the SGPR base
__device__ __forceinline__ void glds16_s_plain(const void *sbase, uint32_t voff, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}

// SGPR base + 32-bit thread offset + immediate byte offset (13-bit signed: -4096 .. 4095).  The instruction offset is
// added on BOTH sides: global address = sbase + voff + OFF, LDS address = M0 + OFF + lane * 16 (measured: a request with
// M0 = slot 3 and offset 3072 left slot 3 untouched) — pass `lds_dst` WITHOUT the offset.
template <int OFF>
__device__ __forceinline__ void glds16_si(const void *sbase, uint32_t voff, uint32_t lds_dst)
{
    static_assert(OFF >= -4096 && OFF <= 4095, "13-bit signed instruction offset");
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "i"(OFF)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

}  // namespace idsp
