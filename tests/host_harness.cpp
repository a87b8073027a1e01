// TEST INFRASTRUCTURE ONLY -- never linked into libkyberhip.so.
// Compiles the device arithmetic headers (kyber_amd/csrc/*.cuh) for the host CPU with g++, so the
// exact field / tower / curve / pairing code the HIP kernels run per lane can be diffed against
// the big-integer oracle in this GPU-less container (tests/test_host_harness.py).
#include <stdint.h>
#include <string.h>

#include "../kyber_amd/csrc/bls12381.cuh"
#include "../kyber_amd/csrc/bls12381_h2c.cuh"
#include "../kyber_amd/csrc/bn256.cuh"
#include "../kyber_amd/csrc/ed25519_h2c.cuh"

using namespace kyb;

extern "C" {

// canonical big-endian in/out through the Montgomery domain
void hh_bls_fp_op(int op, const uint8_t* a48, const uint8_t* b48, uint8_t* out48) {
    uint32_t wa[12], wb[12], wo[12];
    words_from_be<12>(wa, a48);
    words_from_be<12>(wb, b48);
    bls::fp a, b, r;
    fp_from_words<bls::FC>(a, wa);
    fp_from_words<bls::FC>(b, wb);
    switch (op) {
        case 0: fp_mul(r, a, b); break;
        case 1: fp_add(r, a, b); break;
        case 2: fp_sub(r, a, b); break;
        case 3: fp_neg(r, a); break;
        case 4: fp_inv(r, a); break;
        default: fp_sqr(r, a); break;
    }
    fp_to_words<bls::FC>(wo, r);
    words_to_be<12>(out48, wo);
}
int hh_bls_g1_decode(const uint8_t* in, int check) { bls::g1_aff a; return bls::g1_decode(a, in, check != 0); }
int hh_bls_g2_decode(const uint8_t* in, int check) { bls::g2_aff a; return bls::g2_decode(a, in, check != 0); }
int hh_bls_g1_recode(const uint8_t* in, uint8_t* out) {
    bls::g1_aff a;
    int st = bls::g1_decode(a, in, false);
    bls::g1_encode(out, a);
    return st;
}
int hh_bls_g2_recode(const uint8_t* in, uint8_t* out) {
    bls::g2_aff a;
    int st = bls::g2_decode(a, in, false);
    bls::g2_encode(out, a);
    return st;
}
int hh_bls_g1_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out) { return bls::g1_mul_wire(out, k, pt); }
int hh_bls_g2_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out) { return bls::g2_mul_wire(out, k, pt); }
int hh_bls_pair(const uint8_t* g1, const uint8_t* g2, uint8_t* gt) { return bls::pair_wire(gt, g1, g2); }
int hh_bls_pair_check(const uint8_t* p1, const uint8_t* p2, const uint8_t* i1, const uint8_t* i2, uint8_t* ok) {
    return bls::pair_check_wire(ok, p1, p2, i1, i2);
}

// flag-aware variants (KYB_F_UNCOMPRESSED / _OUT / TRUSTED): ints come before the output buffers
int hh_bls_g1_mul_f(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) {
    return bls::g1_mul_wire(out, k, pt, (uint32_t)flags);
}
int hh_bls_g2_mul_f(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) {
    return bls::g2_mul_wire(out, k, pt, (uint32_t)flags);
}
int hh_bls_pair_f(const uint8_t* g1, const uint8_t* g2, int flags, uint8_t* gt) {
    return bls::pair_wire(gt, g1, g2, (uint32_t)flags);
}
int hh_bls_pair_check_f(const uint8_t* p1, const uint8_t* p2, const uint8_t* i1, const uint8_t* i2, int flags,
                        uint8_t* ok) {
    return bls::pair_check_wire(ok, p1, p2, i1, i2, (uint32_t)flags);
}
int hh_bls_g1_unmarshal(const uint8_t* in, int flags, uint8_t* out) { return bls::g1_unmarshal_wire(out, in, (uint32_t)flags); }
int hh_bls_g2_unmarshal(const uint8_t* in, int flags, uint8_t* out) { return bls::g2_unmarshal_wire(out, in, (uint32_t)flags); }
int hh_bls_g1_decode_unc(const uint8_t* in, int validate) { bls::g1_aff a; return bls::g1_decode_unc(a, in, validate != 0); }
int hh_bls_g2_decode_unc(const uint8_t* in, int validate) { bls::g2_aff a; return bls::g2_decode_unc(a, in, validate != 0); }

// Fp12 operations on GT-encoded operands (576 bytes, coefficients < p): the tower arithmetic at chosen magnitudes
// (all coefficients p - 1 drives the lazy Karatsuba sums to their 2p / 4p / 8p bounds)
int hh_bls_fp12_op(int op, const uint8_t* a576, const uint8_t* b576, uint8_t* out576) {
    bls::fp12 a, b, r;
    int st = bls::gt_decode(a, a576);
    if (st) return st;
    st = bls::gt_decode(b, b576);
    if (st) return st;
    switch (op) {
        case 0: fp12_mul(r, a, b); break;
        case 1: fp12_sqr(r, a); break;
        case 2: fp12_cyclo_sqr(r, a); break;
        case 3: fp12_cyclo_sqr_n(r, a, 5); break;
        default: r = a; fp12_mul_by_014(r, b.c0.c0, b.c0.c1, b.c1.c1); break;
    }
    bls::gt_encode(out576, r);
    return 0;
}

// ---- bn256
void hh_bn_fp_op(int op, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    bn::fp a, b, r;
    bn::fp_decode(a, a32);
    bn::fp_decode(b, b32);
    switch (op) {
        case 0: fp_mul(r, a, b); break;
        case 1: fp_add(r, a, b); break;
        case 2: fp_sub(r, a, b); break;
        case 3: fp_neg(r, a); break;
        case 4: fp_inv(r, a); break;
        default: fp_sqr(r, a); break;
    }
    bn::fp_encode(out32, r);
}
int hh_bn_g1_decode(const uint8_t* in) { bn::g1_aff a; return bn::g1_decode(a, in); }
int hh_bn_g2_decode(const uint8_t* in) { bn::g2_aff a; return bn::g2_decode(a, in); }
int hh_bn_g1_unmarshal(const uint8_t* in, uint8_t* out) { return bn::g1_unmarshal_wire(out, in); }
int hh_bn_g2_unmarshal(const uint8_t* in, uint8_t* out) { return bn::g2_unmarshal_wire(out, in); }
int hh_bn_g1_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out) { return bn::g1_mul_wire(out, k, pt); }
int hh_bn_g2_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out) { return bn::g2_mul_wire(out, k, pt); }
int hh_bn_pair(const uint8_t* g1, const uint8_t* g2, uint8_t* gt) { return bn::pair_wire(gt, g1, g2); }
int hh_bn_pair_check(const uint8_t* p1, const uint8_t* p2, const uint8_t* i1, const uint8_t* i2, uint8_t* ok) {
    return bn::pair_check_wire(ok, p1, p2, i1, i2);
}
// Fp12 operations on GT-encoded operands (384 bytes): the shared tower code at bn256's parameters (two lazy levels)
int hh_bn_fp12_op(int op, const uint8_t* a384, const uint8_t* b384, uint8_t* out384) {
    bn::fp12 a, b, r;
    bn::gt_decode(a, a384);
    bn::gt_decode(b, b384);
    switch (op) {
        case 0: fp12_mul(r, a, b); break;
        case 1: fp12_sqr(r, a); break;
        case 2: fp12_cyclo_sqr(r, a); break;
        default: fp12_cyclo_sqr_n(r, a, 5); break;
    }
    bn::gt_encode(out384, r);
    return 0;
}
int hh_bn_gt_mul(const uint8_t* k, const uint8_t* gt, uint8_t* out) { return bn::gt_mul_wire(out, k, gt); }
int hh_bls_gt_mul(const uint8_t* k, const uint8_t* gt, uint8_t* out) { return bls::gt_mul_wire(out, k, gt); }
int hh_bn_hash_g1(const uint8_t* msg, int len, uint8_t* out) { return bn::hash_g1_wire(out, msg, (size_t)len); }
static bls::DstArg mk_dst(const uint8_t* dst, int len) {
    bls::DstArg d;
    memset(&d, 0, sizeof d);
    memcpy(d.b, dst, (size_t)len);
    d.len = (uint32_t)len;
    return d;
}
int hh_bls_hash_g1(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bls::hash_g1_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
int hh_bls_hash_g2(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bls::hash_g2_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
int hh_bls_verify_g2(const uint8_t* pk, const uint8_t* msg, int len, const uint8_t* dst, int dlen, const uint8_t* sig,
                     uint8_t* ok) {
    return bls::verify_g2_wire(ok, pk, msg, (size_t)len, mk_dst(dst, dlen), sig);
}
int hh_bls_verify_g1(const uint8_t* pk, const uint8_t* msg, int len, const uint8_t* dst, int dlen, const uint8_t* sig,
                     uint8_t* ok) {
    return bls::verify_g1_wire(ok, pk, msg, (size_t)len, mk_dst(dst, dlen), sig);
}
void hh_ed_hash(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    EdDstArg d;
    memset(&d, 0, sizeof d);
    memcpy(d.b, dst, (size_t)dlen);
    d.len = (uint32_t)dlen;
    ed_hash_wire(out, msg, (size_t)len, d);
}
}
