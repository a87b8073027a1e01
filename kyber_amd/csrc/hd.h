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
// host build = tests/host_harness.cpp, one translation unit: the out-of-line boundaries mean nothing there, and gcc
// warns about every function that is both `inline` and `noinline`
#define KYB_HD_NOINLINE inline
#endif

// Register budget (waves per SIMD) of the kernels of a translation unit that name no budget of their own.  Out-of-line
// device functions take the LOOSEST budget of the kernels that reach them, so one kernel without a budget lets every
// shared callee -- and with it every kernel of the unit -- grow to 512 registers: a unit that wants two waves per SIMD
// says so for all of its kernels (-DKYB_TU_WAVES=2).
#ifndef KYB_TU_WAVES
#define KYB_TU_WAVES 1
#endif

// Per-call flags of the pairing-suite entry points (values mirror include/kyber_hip.h; context.hip static_asserts).
namespace kyb {
constexpr uint32_t FLAG_UNCOMPRESSED = 2u;  // BLS12-381 point inputs are ZCash uncompressed (96 / 192 B)
constexpr uint32_t FLAG_UNCOMPRESSED_OUT = 4u;  // BLS12-381 point outputs of mul are uncompressed too
constexpr uint32_t FLAG_TRUSTED0 = 0x100u;  // point argument i (bit 8 + i) was validated before: skip its checks
KYB_HD bool flag_trusted(uint32_t flags, int arg) { return (flags >> (8 + arg)) & 1u; }

struct DstArg {  // hash-to-curve domain separation tag, passed by value to the kernels (RFC 9380: at most 255 bytes)
    uint8_t b[256];
    uint32_t len;
};
}  // namespace kyb
