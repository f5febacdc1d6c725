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
