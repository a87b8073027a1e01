// Does ONE wave per SIMD keep the VALU busy on Montgomery multiplications?  Each lane runs a dependent chain of
// fp_mul (BLS12-381, mont.cuh) -- ILP inside a multiplication only -- or ILP independent chains interleaved.  The same
// work is launched as 1024 waves (one per SIMD) and as 2048 / 4096 waves (registers are few, so they co-reside):
// if time(2048) ~ 2 x time(1024) a lone wave already saturates its SIMD; if it is well below 2 x, a lone wave stalls
// on instruction dependencies and the pairing kernels (one wave per SIMD) inherit that.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ikyber_amd/csrc -Iinclude tools/fpmul_probe.hip -o gpurun_out/fpmul_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "bls12381.cuh"

using namespace kyb;
using namespace kyb::bls;

template <int ILP, bool SQR>
__global__ __launch_bounds__(64) void probe(uint32_t* out, int iters, uint32_t salt) {
    fp x[ILP], y;
#pragma unroll
    for (int l = 0; l < FC::NWORDS; l++) y.v[l] = (threadIdx.x * 2654435761u + l * 40503u + salt) & 0x0FFFFFFFu;
#pragma unroll
    for (int k = 0; k < ILP; k++)
#pragma unroll
        for (int l = 0; l < FC::NWORDS; l++) x[k].v[l] = (threadIdx.x * 40503u + blockIdx.x * 97u + k * 13u + l + salt) & 0x0FFFFFFFu;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            if (SQR) fp_sqr(x[k], x[k]);
            else fp_mul(x[k], x[k], y);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++)
#pragma unroll
        for (int l = 0; l < FC::NWORDS; l++) s ^= x[k].v[l];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, bool SQR>
static float run(int waves, int iters) {
    uint32_t* d;
    hipMalloc(&d, (size_t)waves * 64 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((probe<ILP, SQR>), dim3(waves), dim3(64), 0, 0, d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<ILP, SQR>), dim3(waves), dim3(64), 0, 0, d, iters, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms;
}

template <int ILP, bool SQR>
static void report(const char* name) {
    const int iters = 4096 / ILP;
    float t[3];
    const int w[3] = {1024, 2048, 4096};
    for (int i = 0; i < 3; i++) t[i] = run<ILP, SQR>(w[i], iters);
    // cycles per multiplication per wave at one wave per SIMD (2.4 GHz)
    printf("{\"op\": \"%s\", \"ilp\": %d, \"ms_1024\": %.3f, \"ms_2048\": %.3f, \"ms_4096\": %.3f, \"cycles_per_op_1wave\": %.0f, "
           "\"ratio_2048\": %.2f, \"ratio_4096\": %.2f}\n",
           name, ILP, t[0], t[1], t[2], t[0] * 1e-3 * 2.4e9 / (iters * ILP), t[1] / t[0], t[2] / t[0]);
}

int main() {
    report<1, false>("fp_mul");
    report<2, false>("fp_mul");
    report<3, false>("fp_mul");
    report<1, true>("fp_sqr");
    report<2, true>("fp_sqr");
    return 0;
}
