// Occupancy probe: does a kernel that uses private scratch get a second wave per SIMD on MI355X?
// Each lane chases a pointer chain through an L2-resident table (latency-bound: a lone wave leaves its SIMD idle;
// dependent ALU chains would not do -- the 4-cycle cadence lets one wave saturate the VALU), holding ~200 VGPRs so
// that at most 2 waves fit a SIMD.  If 2048 waves finish in about the time of 1024, waves co-reside;
// if the time doubles they ran one per SIMD in two rounds.  Variant 1 adds a dynamically indexed private array
// (forces scratch).  build: hipcc -O3 --offload-arch=gfx950 tools/occ_probe.hip -o gpurun_out/occ_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int SCRATCH>
__global__ __launch_bounds__(64) void probe(uint64_t* out, const uint32_t* __restrict__ chain, int iters, int salt) {
    uint64_t acc[96];  // 192 VGPRs of live state
#pragma unroll
    for (int i = 0; i < 96; i++) acc[i] = threadIdx.x * 1315423911ull + i * 2654435761ull + salt;
    uint32_t priv[SCRATCH ? 256 : 1];
    if (SCRATCH) {
        for (int i = 0; i < 256; i++) priv[i] = i * 7 + salt;
    }
    uint64_t x = acc[0];
    uint32_t idx = (blockIdx.x * 13) & 1023;  // wave-uniform row: every load is one coalesced 256-byte row
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        idx = chain[idx * 64 + threadIdx.x];  // dependent load: a round trip with nothing else to do meanwhile
        x += idx;
        acc[it % 96 == 0 ? 0 : 1] ^= x;
        if (SCRATCH) x += priv[(x >> 3) & 255];  // lane-divergent index: private memory, not registers
    }
    uint64_t s = x;
#pragma unroll
    for (int i = 0; i < 96; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int S>
static float run(int waves, int iters) {
    uint64_t* d;
    hipMalloc(&d, (size_t)waves * 64 * 8);
    static uint32_t* chain = nullptr;
    if (!chain) {
        uint32_t* h = (uint32_t*)malloc(65536 * 4);
        for (uint32_t i = 0; i < 65536; i++) h[i] = ((i / 64) * 397u + 123u) & 1023u;  // row -> next row
        hipMalloc(&chain, 65536 * 4);
        hipMemcpy(chain, h, 65536 * 4, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<S>, dim3(waves), dim3(64), 0, 0, d, chain, iters, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<S>, dim3(waves), dim3(64), 0, 0, d, chain, iters, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms;
}

int main() {
    const int iters = 2000;
    printf("{");
    for (int waves = 1024; waves <= 4096; waves *= 2)
        printf("\"regs_only_%d_waves_ms\": %.3f, \"with_scratch_%d_waves_ms\": %.3f%s", waves, run<0>(waves, iters), waves,
               run<1>(waves, iters), waves < 4096 ? ", " : "");
    printf("}\n");
    return 0;
}
