// GPU check of the lazy limb ladders against the packed code (debugging aid, round 5)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../kyber_amd/csrc/bn256.cuh"
using namespace kyb;
using L = Limb30<Bn256Fp>;
__device__ void put(uint32_t* out, int row, const bn::g1_jac& j) {
    bn::g1_aff a;
    jac_to_aff(a, j);
    for (int w = 0; w < 8; w++) { out[row * 16 + w] = a.x.v[w]; out[row * 16 + 8 + w] = a.y.v[w]; }
}
__global__ void k(uint32_t* out) {
    bn::g1_jac p, d0, t;
    for (int j = 0; j < 8; j++) { p.X.v[j] = Bn256Curve::G1X[j]; p.Y.v[j] = Bn256Curve::G1Y[j]; }
    fp_one(p.Z);
    jac_dbl(d0, p);  // 2P packed
    put(out, 0, d0);
    JacLz<bn::lz1> q;
    bn::lz1::enter(q.X, p.X); bn::lz1::enter(q.Y, p.Y); bn::lz1::one(q.Z); q.inf = 0;
    jaclz_dbl(q);
    jaclz_leave(t, q);
    put(out, 1, t);       // 2P lazy
    jac_madd(d0, d0, p.X, p.Y, false);  // 3P packed
    put(out, 2, d0);
    typename bn::lz1::E x2, y2;
    bn::lz1::enter(x2, p.X); bn::lz1::enter(y2, p.Y);
    jaclz_madd(q, x2, y2, false);
    jaclz_leave(t, q);
    put(out, 3, t);       // 3P lazy
    jaclz_dbl(q); jaclz_dbl(q); jaclz_madd(q, x2, y2, true);  // 11P
    jaclz_leave(t, q);
    put(out, 4, t);
    jac_dbl(d0, d0); jac_dbl(d0, d0);
    bn::fp ny; fp_neg(ny, p.Y);
    jac_madd(d0, d0, p.X, ny, false);
    put(out, 5, d0);
    uint32_t kk[8] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x11111111u, 0x22222222u, 0x33333333u, 0x04444444u};
    bn::g1_mul_glv(t, p, kk);
    put(out, 6, t);
    bn::g1_mul_glv_lz(t, p, kk);
    put(out, 7, t);
}
int main() {
    uint32_t* d; hipMalloc(&d, 128 * 4); uint32_t h[128];
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d); hipMemcpy(h, d, 128 * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"2P packed", "2P lazy  ", "3P packed", "3P lazy  ", "11P lazy ", "11P packd", "kP packed", "kP lazy  "};
    for (int r = 0; r < 8; r++) { printf("%s ", names[r]); for (int j = 15; j >= 0; j--) printf("%08x", h[r * 16 + j]); printf("\n"); }
    return 0;
}
