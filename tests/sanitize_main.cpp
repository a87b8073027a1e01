// AddressSanitizer + UndefinedBehaviorSanitizer run of the device headers' host build (tests/host_harness.cpp) -- test
// infrastructure: tests/test_host_harness_sanitizers.py compiles this file with -fsanitize=address,undefined, feeds it a
// script of calls on stdin and compares what it prints with the plain library's answers.
//   line:  <function> <arg> <arg> ...     arg = x<hex bytes> (input buffer) | i<int> | o<size> (output buffer)
//   reply: <return code> <hex of every output buffer, in order>
#include "host_harness.cpp"

#include <stdio.h>

#include <string>
#include <vector>

typedef int (*fn_t)(...);
struct Named {
    const char* name;
    void* fn;
};
#define F(x) {#x, (void*)&x}
static const Named TABLE[] = {F(hh_bls_g1_fb_mul), F(hh_bls_g2_fb_mul), F(hh_bn_g1_fb_mul), F(hh_bn_g2_fb_mul), F(hh_scalar_poly_eval),
                              F(hh_bls_g1_mul), F(hh_bls_g2_mul), F(hh_bn_g1_mul), F(hh_bn_g2_mul), F(hh_bn4_g1_mul), F(hh_ed_mul),
                              F(hh_bls_g1_decode), F(hh_bls_g2_decode), F(hh_bls_g1_unmarshal), F(hh_bls_g2_unmarshal),
                              F(hh_bls_g1_coop), F(hh_bls_g2_coop), F(hh_bn_g1_coop), F(hh_bls_hash_g1), F(hh_ed_hash),
                              F(hh_bls_g1_xyzz_sum), F(hh_bn_g2_xyzz_sum), F(hh_bls_g1_table8)};

int main() {
    char* line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, stdin) > 0) {
        std::vector<std::string> tok;
        for (char* p = strtok(line, " \n"); p; p = strtok(nullptr, " \n")) tok.push_back(p);
        if (tok.empty()) continue;
        void* fn = nullptr;
        for (const Named& n : TABLE)
            if (tok[0] == n.name) fn = n.fn;
        if (!fn) {
            printf("unknown %s\n", tok[0].c_str());
            return 2;
        }
        std::vector<std::vector<uint8_t>> bufs(tok.size());
        std::vector<int> outs;
        uintptr_t a[8] = {0};
        int na = 0;
        for (size_t i = 1; i < tok.size() && na < 8; i++) {
            const std::string& t = tok[i];
            if (t[0] == 'i') {
                a[na++] = (uintptr_t)(intptr_t)atoi(t.c_str() + 1);
            } else if (t[0] == 'x') {
                for (size_t k = 1; k + 1 < t.size(); k += 2) bufs[i].push_back((uint8_t)strtol(t.substr(k, 2).c_str(), nullptr, 16));
                bufs[i].push_back(0);  // (an empty buffer still has an address)
                a[na++] = (uintptr_t)bufs[i].data();
            } else {
                bufs[i].assign((size_t)atoi(t.c_str() + 1), 0);
                outs.push_back((int)i);
                a[na++] = (uintptr_t)bufs[i].data();
            }
        }
        typedef int (*f8)(uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t);
        const int rc = ((f8)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
        printf("%d", rc);
        for (int i : outs) {
            printf(" ");
            for (uint8_t b : bufs[i]) printf("%02x", b);
        }
        printf("\n");
    }
    return 0;
}
