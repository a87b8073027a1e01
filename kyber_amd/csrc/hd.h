// Host/device portability macros for the arithmetic headers (mont.cuh, tower.cuh, curve.cuh ...).
//
// The product only ever compiles these headers with hipcc for gfx950.  The same headers also
// compile with plain g++ so that tests/host_harness.cpp can run the *identical* field / tower /
// pairing code on the CPU and diff it against the big-integer oracle while debugging -- that
// harness is test infrastructure and is never linked into libkyberhip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KYB_HD __host__ __device__ __forceinline__
#define KYB_HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define KYB_HD inline
#define KYB_HD_NOINLINE __attribute__((noinline)) inline
#endif
