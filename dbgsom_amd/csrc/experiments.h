// Experiment builds only (tools/build_variant.sh <name> <source> "-DDBGSOM_EXPERIMENTS -D..."): in-kernel stamps and
// switches that show what a kernel's time is made of.  Never part of libdbgsom_hip.so; results of such builds are
// timings, sometimes with wrong numbers in the outputs the stamps are carried in.
#pragma once

// ---- segsum_chain_kernel (accumulate.hip) -------------------------------------------------------------------
// CHAIN_STAMPS=1: per-phase s_memtime sums of a workgroup (thread 0's view), left in `dist` of the chunk's first
// rows as -(1e12 + phase 1e10 + cycles); tools/chain_stamps.py decodes them.  CHAIN_PAD: bytes of LDS a workgroup
// asks for on top (fewer workgroups per CU).  CHAIN_CB / CHAIN_CW / CHAIN_OCC: see accumulate.hip.
#ifndef CHAIN_STAMPS
#define CHAIN_STAMPS 0
#endif
#ifndef CHAIN_PAD
#define CHAIN_PAD 0
#endif
#if CHAIN_STAMPS
#define CSTAMP_DECL uint64_t st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime()
#define CSTAMP(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); if (tid == 0) { st_acc[k] += now_ - st_last; } st_last = now_; } while (0)
#define CSTAMP_LOADS_LANDED do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CSTAMP(2); } while (0)
#define CSTAMP_FLUSH(dist, rows_s, n)                                                                          \
    do {                                                                                                       \
        CSTAMP(7);                                                                                             \
        if (tid == 0 && (n) >= 11) {                                                                           \
            for (int k_ = 0; k_ < 10; ++k_) (dist)[(rows_s)[k_]] = -(1e12 + 1e10 * k_ + (double)st_acc[k_]);     \
            (dist)[(rows_s)[10]] = -(1e12 + 1e10 * 10 + (double)(n));                                           \
        }                                                                                                      \
    } while (0)
#else
#define CSTAMP_DECL
#define CSTAMP(k)
#define CSTAMP_LOADS_LANDED
#define CSTAMP_FLUSH(dist, rows_s, n)
#endif

// ---- subset_exact_kernel (filter.hip) -----------------------------------------------------------------------
// EXACT_TIMELINE=1: a workgroup's life in s_memrealtime ticks (100 MHz, one clock for all XCDs) and where it ran, carried as NEGATIVE distances
// of its first six samples: -(1 + (class << 51 | workgroup << 33 | (value & 0x3fffffff) << 3 | field)), fields: start, end,
// list length << 8 | class << 4, XCC << 16 | HW_ID, first tile landed, last distances pushed, s_memtime (shader clock) at start and end -- tools/exact_timeline.py.
#ifndef EXACT_TIMELINE
#define EXACT_TIMELINE 0
#endif
#if EXACT_TIMELINE
#define XT_DECL uint64_t xt_v[8] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0, __builtin_amdgcn_s_memtime(), 0}
#define XT_MARK(k) do { xt_v[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XT_MARK_ONCE(k) do { if (xt_v[k] == 0) xt_v[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XT_FLUSH(JTL_, cnt_, dist_, isamp0_, Kk_)                                                              \
    do {                                                                                                       \
        xt_v[7] = __builtin_amdgcn_s_memtime();                                                                \
        xt_v[2] = ((uint64_t)(cnt_) << 8) | ((uint64_t)(JTL_) << 4);                                           \
        xt_v[3] = ((uint64_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) |                                 \
                  ((uint64_t)__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffff);                               \
        if (wave == 0 && lq == 0 && lr < 8 && (isamp0_) >= 0) {                                                \
            const uint64_t key_ = ((uint64_t)(JTL_) << 18) | (blockIdx.x & 0x3ffff);                           \
            uint64_t val_ = xt_v[0];                                                                           \
            for (int f_ = 1; f_ < 8; ++f_) if (lr == f_) val_ = xt_v[f_];                                      \
            (dist_)[(isamp0_) * (Kk_)] = -(double)((key_ << 33) | ((val_ & 0x3fffffffull) << 3) | (uint64_t)lr) - 1.0; \
        }                                                                                                      \
    } while (0)
#else
#define XT_DECL
#define XT_MARK(k)
#define XT_MARK_ONCE(k)
#define XT_FLUSH(JTL_, cnt_, dist_, isamp0_, Kk_)
#endif

// ---- prune_mark_kernel (filter.hip) -------------------------------------------------------------------------
// PRUNE_STAMPS=1: s_memrealtime (100 MHz) of thread 0 at the kernel's phase boundaries, 8 values per workgroup in
// a device array; dbgsom_experiment_prune_stamps() copies it out (tools/prune_timeline.py).
#ifndef PRUNE_STAMPS
#define PRUNE_STAMPS 0
#endif
#if PRUNE_STAMPS
__device__ unsigned long long g_prune_stamps[8 * 16384];
#define PM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) { g_prune_stamps[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); if ((k) == 0) g_prune_stamps[8 * blockIdx.x + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) | ((unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffff); } } while (0)
#else
#define PM_STAMP(k)
#endif
