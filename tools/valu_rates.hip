// Issue cost of the individual VALU instructions the field kernels are made of, gfx950, by inline assembly so that
// the instruction measured is the instruction named (tools/valu_peak.hip leaves the selection to the compiler: its
// "mul_lo_u32+add" line is a v_mad_u64_u32).  Each kernel runs CHAINS independent dependency chains of one
// instruction per wave, 4 waves per SIMD; prints cycles per wave64 instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 2048
#define CHAINS 8

#define RATE_KERNEL(NAME, DECL, INIT, BODY, FOLD)                                     \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint32_t seed) {       \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;                   \
        (void)a; (void)b;                                                             \
        DECL;                                                                         \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { INIT; }                  \
        for (int it = 0; it < ITERS; it++) {                                          \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY; }              \
        }                                                                             \
        uint64_t r = 0;                                                               \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { FOLD; }                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                               \
    }

RATE_KERNEL(k_add_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_and_b32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_lshl_add_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_mul_lo_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_mul_i32_i24, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_mul_hi_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_cndmask, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(b) : "vcc"), r += x[c])
RATE_KERNEL(k_mad_i64_i32, uint64_t x[CHAINS], x[c] = a + c, asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc"), r += x[c])
RATE_KERNEL(k_mad_u64_u32, uint64_t x[CHAINS], x[c] = a + c, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc"), r += x[c])
RATE_KERNEL(k_ashr_i64, uint64_t x[CHAINS], x[c] = ((uint64_t)a << 32) + c, asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(x[c])), r += x[c])
RATE_KERNEL(k_lshl_add_u64, uint64_t x[CHAINS]; uint64_t y = ((uint64_t)b << 32) | a, x[c] = a + c, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[c]) : "v"(y)), r += x[c])
RATE_KERNEL(k_add_co_pair, uint32_t xl[CHAINS]; uint32_t xh[CHAINS], xl[c] = a + c; xh[c] = b + c,
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(xl[c]), "+v"(xh[c]) : "v"(a), "v"(b) : "vcc"),
            r += xl[c] + xh[c])
RATE_KERNEL(k_alignbit, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_bfe_i32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_bfe_i32 %0, %0, 1, 26" : "+v"(x[c])), r += x[c])
RATE_KERNEL(k_ashr_i32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x[c])), r += x[c])
RATE_KERNEL(k_mad_u32_u24, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)), r += x[c])
RATE_KERNEL(k_add3_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)), r += x[c])
RATE_KERNEL(k_sub_u32, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x[c]) : "v"(b)), r += x[c])
RATE_KERNEL(k_and_or, uint32_t x[CHAINS], x[c] = a + c, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)), r += x[c])
// a MAD and a plain 32-bit operation on another register, alternating: do they overlap?
RATE_KERNEL(k_mad_plus_add, uint64_t x[CHAINS]; uint32_t z[CHAINS], x[c] = a + c; z[c] = b + c,
            asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %3" : "+v"(x[c]), "+v"(z[c]) : "v"(a), "v"(b) : "vcc"), r += x[c] + z[c])

template <class K>
void run(const char* name, K kern, uint64_t* d, int per_body) {
    const int blocks = 256 * 4, threads = 256;  // one 256-thread group per SIMD quadruple: 4 waves per CU x 4 = 16 waves per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double simds = 4.0 * p.multiProcessorCount;
    const double wave_instr_per_simd = (double)reps * blocks * (threads / 64) * (double)ITERS * CHAINS * per_body / simds;
    const double cycles = ms * 1e-3 * (p.clockRate * 1e3);
    printf("{\"instr\": \"%s\", \"cycles_per_wave_instr\": %.3f, \"lane_ops_per_s\": %.4e}\n", name, cycles / wave_instr_per_simd,
           (double)reps * blocks * threads * (double)ITERS * CHAINS * per_body / (ms * 1e-3));
}

int main() {
    uint64_t* d;
    (void)hipMalloc(&d, 256 * 4 * 256 * 8);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
#define RUN(n, per) run(#n, k_##n, d, per)
    RUN(add_u32, 1);
    RUN(sub_u32, 1);
    RUN(and_b32, 1);
    RUN(and_or, 1);
    RUN(add3_u32, 1);
    RUN(lshl_add_u32, 1);
    RUN(ashr_i32, 1);
    RUN(bfe_i32, 1);
    RUN(cndmask, 1);
    RUN(alignbit, 1);
    RUN(mul_lo_u32, 1);
    RUN(mul_i32_i24, 1);
    RUN(mad_u32_u24, 1);
    RUN(mul_hi_u32, 1);
    RUN(mad_i64_i32, 1);
    RUN(mad_u64_u32, 1);
    RUN(ashr_i64, 1);
    RUN(lshl_add_u64, 1);
    RUN(add_co_pair, 2);
    RUN(mad_plus_add, 2);
    return 0;
}
