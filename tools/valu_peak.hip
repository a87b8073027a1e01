// Integer-VALU issue-rate micro-benchmark for gfx950: defines the denominator
// of the "VALU integer multiply" roofline the Ed25519 / pairing kernels are
// bound by.  Build: hipcc --offload-arch=gfx950 -O3
// tools/valu_peak.hip -o gpurun_out/valu_peak ; prints ops/s per instruction kind.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CHAINS];
    double facc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { acc[c] = c + a; facc[c] = (double)(c + a); }
    double fa = (double)a * 1e-9, fb = (double)b * 1e-9;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (KIND == 0) acc[c] = (uint64_t)((uint32_t)acc[c] ^ a) * b + acc[c];             // v_mad_u64_u32 (+xor)
            if (KIND == 1) acc[c] = (uint64_t)((int64_t)(int32_t)((uint32_t)acc[c] ^ a) * (int64_t)(int32_t)b + (int64_t)acc[c]);  // v_mad_i64_i32
            if (KIND == 2) acc[c] = (uint32_t)acc[c] * b + a;                                   // v_mul_lo_u32 (+add) / v_mad_u32?
            if (KIND == 3) acc[c] = __umulhi((uint32_t)acc[c], b) + a;                          // v_mul_hi_u32
            if (KIND == 4) acc[c] = __umul24((uint32_t)acc[c], b) + a;                          // v_mad_u32_u24
            // (KIND 5, a bare v_add_u32 of a loop-invariant, and KIND 7, its 64-bit form, were folded by the compiler into one
            // addition per chain -- their rows claimed 3x the chip's full-rate VALU ceiling; removed in round 6.
            // The loop body of every remaining kind is dumped beside the figures: profiles/r06_valu_peak_isa.txt)
            if (KIND == 6) facc[c] = fma(facc[c], fa, fb);                                      // v_fma_f64
            if (KIND == 8) acc[c] = (uint64_t)((int64_t)acc[c] >> 26) + b;                      // 64-bit ashr + add
            if (KIND == 9) acc[c] = (uint64_t)acc[c] * 3 + ((uint64_t)a * b);                   // pure mad chain (mul by small const + mad)
            if (KIND == 10) acc[c] = (acc[c] << 3) + (((uint64_t)b << 32) | a);                  // v_lshl_add_u64
            if (KIND == 11) acc[c] = (uint64_t)(__builtin_amdgcn_alignbit((uint32_t)(acc[c] >> 32), (uint32_t)acc[c], 26) + a) | ((uint64_t)b << 32);  // alignbit + add
            if (KIND == 12) { uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32); acc[c] = (uint64_t)lo * hi + acc[c]; }  // v_mad_u64_u32 fully dependent
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r += acc[c] + (uint64_t)facc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
double run(const char* name, uint64_t* d) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)reps * blocks * threads * (double)ITERS * CHAINS;
    double rate = ops / (ms * 1e-3);
    printf("{\"kind\": \"%s\", \"lane_ops_per_s\": %.4e, \"ms\": %.3f}\n", name, rate, ms / reps);
    return rate;
}

int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<0>("mad_u64_u32+xor", d);
    run<1>("mad_i64_i32+xor", d);
    run<2>("mul_lo_u32+add", d);
    run<3>("mul_hi_u32+add", d);
    run<4>("mad_u32_u24", d);
    run<6>("fma_f64", d);
    run<8>("ashr_i64+add", d);
    run<10>("lshl_add_u64", d);
    run<11>("alignbit+add(+or)", d);
    run<12>("mad_u64_u32 dependent operands", d);
    return 0;
}
