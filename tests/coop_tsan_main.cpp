// TEST INFRASTRUCTURE ONLY: driver of tests/test_coop_slots_tsan.py -- the cooperative slot arithmetic (coop_slots.cuh) with four
// threads as lanes under ThreadSanitizer: any pair of slot accesses of one level that the barriers do not separate is a
// reported data race (on the GPU it would be a wrong sum that depends on timing).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
extern "C" int hh_bn_g1_coop(const uint8_t* ops, const uint8_t* a, const uint8_t* b, uint8_t* out);
extern "C" int hh_bn_g1_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out);
int main() {
    uint8_t g[64] = {0}, k[32] = {0}, A[64], B[64], out[128];
    g[31] = 1; g[63] = 2;  // (1, 2) on y^2 = x^3 + 3
    k[31] = 7; hh_bn_g1_mul(k, g, A);
    k[31] = 11; hh_bn_g1_mul(k, g, B);
    const char* progs[] = {"a", "d", "adnxa", "as", "ddda"};
    for (const char* p : progs) {
        int st = hh_bn_g1_coop((const uint8_t*)p, A, B, out);
        printf("%s st %d %02x\n", p, st, out[0]);
    }
    // P = Q (fallback) and infinity
    int st = hh_bn_g1_coop((const uint8_t*)"a", A, A, out); printf("eq st %d\n", st);
    uint8_t inf[64] = {0};
    st = hh_bn_g1_coop((const uint8_t*)"a", inf, B, out); printf("inf st %d\n", st);
    return 0;
}
