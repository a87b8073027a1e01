// TEST INFRASTRUCTURE ONLY -- never linked into libkyberhip.so.
// Compiles the device arithmetic headers (kyber_amd/csrc/*.cuh) for the host CPU with g++, so the
// exact field / tower / curve / pairing code the HIP kernels run per lane can be diffed against
// the big-integer oracle in this GPU-less container (tests/test_host_harness.py).
#include <stdint.h>
#include <string.h>

#include "../kyber_amd/csrc/bls12381.cuh"
#include "../kyber_amd/csrc/rowfp.cuh"
#include "../kyber_amd/csrc/bls12381_h2c.cuh"
#include "../kyber_amd/csrc/bn254.cuh"
#include "../kyber_amd/csrc/bn256.cuh"
#include "../kyber_amd/csrc/ed25519_h2c.cuh"
#include "../kyber_amd/csrc/fixed_base.cuh"
#include "../kyber_amd/csrc/bls12381_fb.cuh"
#include "../kyber_amd/csrc/bls12381_keylines.cuh"
#include "../kyber_amd/csrc/bls12381_g1coop.cuh"
#include "../kyber_amd/csrc/coop_slots.cuh"
#include "../kyber_amd/csrc/scalar_field.cuh"
#include <pthread.h>
#include <thread>
#include <vector>

using namespace kyb;


// The MSM's bucket-piece accumulator (curve.cuh Xyzz, msm_ws.cuh piece_madd): sum of +-points through xyzz_madd, left
// through xyzz_to_jac -- held against the oracle's plain sum, exceptional cases included (tests/test_host_harness_*).
template <class F, class AffT, class Dec, class Enc>
static int xyzz_sum(int n, const uint8_t* pts, int wire, const uint8_t* signs, uint8_t* out, Dec dec, Enc enc) {
    Xyzz<F> acc;
    xyzz_set_inf(acc);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        AffT a;
        if (dec(a, pts + (size_t)wire * i)) { bad++; continue; }
        if (a.inf) continue;
        F y = a.y, ny;
        f_neg(ny, a.y);
        f_cmov(y, ny, signs[i] != 0);
        xyzz_madd(acc, a.x, y);
    }
    Jac<F> j;
    xyzz_to_jac(j, acc);
    AffT r;
    jac_to_aff(r, j);
    enc(out, r);
    return bad;
}

// Fixed-base multiplication (fixed_base.cuh): table of the base built on the host under the group's policy P (sub-scalars
// over endomorphism images), nk scalars multiplied through it.  member_out (optional) receives the policy's membership
// verdict read off the finished table.
template <class P, class F, class AffT, class Enc>
static int fb_mul_host(const uint8_t* base, uint32_t flags, int nk, const uint8_t* scalars_be, uint8_t* out, int osz, Enc enc,
                       int* member_out = nullptr) {
    AffT b;
    const int st = P::decode_on_curve(b, base, flags);
    if (st) return st;
    std::vector<fb::Entry<F, P::NI>> tab((size_t)P::NW * fb::NENT);
    if (!b.inf) {
        Jac<F> q[P::NW];
        fb::chain<P::NW>(q, b);
        for (int w = 0; w < P::NW; w++)
            for (int j = 0; j < fb::NENT; j++) fb::entry_images<P>(tab[(size_t)w * fb::NENT + j], q[w], j);
    }
    if (member_out) *member_out = b.inf ? 1 : (P::member(b, tab.data()) ? 1 : 0);
    for (int i = 0; i < nk; i++) {
        uint32_t k[8];
        words_from_be<8>(k, scalars_be + 32 * i);
        AffT a;
        if (b.inf) {
            f_zero(a.x);
            f_zero(a.y);
            a.inf = true;
        } else {
            Jac<F> r;
            fb::mul<P>(r, k, tab.data());
            jac_to_aff(a, r);
        }
        enc(out + (size_t)osz * i, a);
    }
    return 0;
}

// ---- The MSM's cooperative slot arithmetic (coop_slots.cuh) with four threads as the four lanes of a group and a
// pthread barrier as the workgroup barrier.
static pthread_barrier_t g_coop_barrier;
namespace kyb {
void coop_host_sync() { pthread_barrier_wait(&g_coop_barrier); }
}
// ops: a string of steps over three points in slots -- P (0..2) = first input, Q (3..5) = second input:
//   'a' P += Q ; 'n' P += Q computed and NOT committed ; 'd' P = 2 P ; 'x' P = 2 P not committed ; 's' swap roles (Q += P)
template <class F, class AffT, class Dec, class Enc>
static int coop_run(const char* ops, const uint8_t* pa, const uint8_t* pb, int wire, uint8_t* out, Dec dec, Enc enc) {
    AffT a, b;
    if (dec(a, pa) || dec(b, pb)) return 1;
    constexpr int P = 0, Q = 3, T = 6, NS = T + coop::TEMPS;
    static coop::Slot<F> S[NS];
    static uint32_t fl[2];
    Jac<F> ja, jb;
    jac_from_aff(ja, a);
    jac_from_aff(jb, b);
    // scale the inputs to general Jacobian form so that Z != 1 paths are exercised: (X l^2, Y l^3, Z l)
    F l, l2, l3;
    f_one(l);
    f_add(l, l, l);
    f_add(l2, l, l);  // l = 2, l2 = 4 (Montgomery form of small integers via additions of one)
    f_add(l3, l2, l2);
    f_mul(ja.X, ja.X, l2);
    f_mul(ja.Y, ja.Y, l3);
    f_mul(ja.Z, ja.Z, l);
    S[P].f = ja.X; S[P + 1].f = ja.Y; S[P + 2].f = ja.Z;
    S[Q].f = jb.X; S[Q + 1].f = jb.Y; S[Q + 2].f = jb.Z;
    pthread_barrier_init(&g_coop_barrier, nullptr, 4);
    std::thread th[4];
    for (int r = 0; r < 4; r++)
        th[r] = std::thread([=]() {
            for (const char* o = ops; *o; o++) {
                if (*o == 'a') coop::add<F>(S, fl, r, P, Q, T, true);
                else if (*o == 'n') coop::add<F>(S, fl, r, P, Q, T, false);
                else if (*o == 's') coop::add<F>(S, fl, r, Q, P, T, true);
                else if (*o == 'd') coop::dbl<F>(S, r, P, T, true);
                else if (*o == 'x') coop::dbl<F>(S, r, P, T, false);
            }
        });
    for (int r = 0; r < 4; r++) th[r].join();
    pthread_barrier_destroy(&g_coop_barrier);
    for (int which = 0; which < 2; which++) {
        Jac<F> j;
        const int base = which ? Q : P;
        j.X = S[base].f; j.Y = S[base + 1].f; j.Z = S[base + 2].f;
        AffT r;
        jac_to_aff(r, j);
        enc(out + (size_t)wire * which, r);
    }
    return 0;
}

// The same run through the LIMB-form accumulator (curve.cuh XyzzL, fp_limbs.cuh): what accumulate_kernel and the
// fixed-base walks run for base fields with headroom.  `maxk` receives the largest value / p met in X (the lazy bound
// the formulas are proved for is 8).
static int xyzzl_sum_bls_g1(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out, int* max_top) {
    XyzzL<bls::FC> acc;
    xyzzl_set_inf(acc);
    int bad = 0;
    uint32_t top = 0;
    for (int i = 0; i < n; i++) {
        bls::g1_aff a;
        if (bls::g1_decode(a, pts + (size_t)48 * i, false)) { bad++; continue; }
        if (a.inf) continue;
        xyzzl_madd(acc, a.x, a.y, signs[i] != 0);
        for (int j = 0; j < bls::FC::N - 1; j++)
            if (acc.X.l[j] >> bls::FC::W || acc.Y.l[j] >> bls::FC::W || acc.ZZ.l[j] >> bls::FC::W || acc.ZZZ.l[j] >> bls::FC::W) bad += 1000;
        if (acc.X.l[bls::FC::N - 1] > top) top = acc.X.l[bls::FC::N - 1];
    }
    if (max_top) *max_top = (int)top;
    Jac<bls::fp> j;
    xyzzl_to_jac(j, acc);
    bls::g1_aff r;
    jac_to_aff(r, j);
    bls::g1_encode(out, r);
    return bad;
}

// The lazy Jacobian formulas (jac_lazy.cuh) driven directly: ops[i] = 'a' (acc += pts[i]), 's' (acc -= pts[i]) or 'd'
// (acc = 2 acc; pts[i] unused) -- the exceptional branches of the mixed addition (accumulator at infinity, the same point,
// the opposite point) are reached by construction, which a scalar multiplication only does by accident.
template <class LF, bool TIGHT, class F, class AffT, class Dec, class Enc>
static int lz_chain(int n, const char* ops, const uint8_t* pts, int wire, uint8_t* out, Dec dec, Enc enc) {
    JacLz<LF> acc;
    jaclz_set_inf(acc);
    for (int i = 0; i < n; i++) {
        if (ops[i] == 'd') {
            if (!acc.inf) {
                if constexpr (TIGHT) jaclz_dbl_t(acc);
                else jaclz_dbl(acc);
            }
            continue;
        }
        AffT a;
        if (dec(a, pts + (size_t)wire * i)) return 1 + i;
        if (a.inf) continue;
        typename LF::E x, y;
        LF::enter(x, a.x);
        LF::enter(y, a.y);
        if constexpr (TIGHT) jaclz_madd_t(acc, x, y, ops[i] == 's');
        else jaclz_madd(acc, x, y, ops[i] == 's');
    }
    Jac<F> j;
    jaclz_leave(j, acc);
    AffT r;
    jac_to_aff(r, j);
    enc(out, r);
    return 0;
}

extern "C" {

// which: 0 bn256 G1 (LzFp over Limb30), 1 bn256 G2 (LzFp2), 2 BLS12-381 G1 on the native limbs (the _t formulas),
// 3 BLS12-381 G1 on fourteen limbs (generic formulas), 4 BLS12-381 G2 (LzFp2 over fourteen limbs)
int hh_lz_chain(int which, int n, const char* ops, const uint8_t* pts, uint8_t* out) {
    switch (which) {
        case 0:
            return lz_chain<bn::lz1, false, bn::fp, bn::g1_aff>(n, ops, pts, 64, out, [](bn::g1_aff& a, const uint8_t* in) { return bn::g1_decode(a, in); },
                                                                  [](uint8_t* o, const bn::g1_aff& a) { bn::g1_encode(o, a); });
        case 1:
            return lz_chain<bn::lz2, false, bn::fp2, bn::g2_aff>(n, ops, pts, 128, out, [](bn::g2_aff& a, const uint8_t* in) { return bn::g2_decode(a, in); },
                                                                   [](uint8_t* o, const bn::g2_aff& a) { bn::g2_encode(o, a); });
        case 2:
            return lz_chain<LzFpN<bls::FC>, true, bls::fp, bls::g1_aff>(n, ops, pts, 48, out, [](bls::g1_aff& a, const uint8_t* in) { return bls::g1_decode(a, in, false); },
                                                                          [](uint8_t* o, const bls::g1_aff& a) { bls::g1_encode(o, a); });
        case 3:
            return lz_chain<LzFp<Limb30<bls::FC>>, false, bls::fp, bls::g1_aff>(n, ops, pts, 48, out, [](bls::g1_aff& a, const uint8_t* in) { return bls::g1_decode(a, in, false); },
                                                                                  [](uint8_t* o, const bls::g1_aff& a) { bls::g1_encode(o, a); });
        default:
            return lz_chain<LzFp2<Limb30<bls::FC>, bls::TC>, false, bls::fp2, bls::g2_aff>(n, ops, pts, 96, out, [](bls::g2_aff& a, const uint8_t* in) { return bls::g2_decode(a, in, false); },
                                                                                              [](uint8_t* o, const bls::g2_aff& a) { bls::g2_encode(o, a); });
    }
}

int hh_bls_g1_xyzzl_sum(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out, int* max_top) {
    return xyzzl_sum_bls_g1(n, pts, signs, out, max_top);
}
// lazy-form primitives against plain integers: op 0: a - b + 8p, 1: +-a - b + 4p (sign = c[0]), 2: a - b - 2c + 6p,
// 3: (a b + c d) / R, 4: is a == 0 mod p (a = k p + delta).  Inputs are N little-endian limbs (uint32) each.
int hh_bls_fpl_op(int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
    FpL<bls::FC> A, B, Cc, D, R;
    for (int j = 0; j < bls::FC::N; j++) { A.l[j] = a[j]; B.l[j] = b[j]; Cc.l[j] = c[j]; D.l[j] = d[j]; R.l[j] = 0; }
    int rv = 0;
    switch (op) {
        case 0: fpl_sub<8>(R, A, B); break;
        case 1: fpl_sub_signed<4>(R, A, c[0] != 0, B); break;
        case 2: fpl_add_2x(D, B, Cc); fpl_sub<6>(R, A, D); break;
        case 3: fpl_mul2sum(R, A, B, Cc, D); break;
        default: rv = fpl_is_zero_mod_p<9>(A) ? 1 : 0; break;
    }
    for (int j = 0; j < bls::FC::N; j++) out[j] = R.l[j];
    return rv;
}

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

int hh_bls_g1_xyzz_sum(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out) {
    return xyzz_sum<bls::fp, bls::g1_aff>(n, pts, 48, signs, out,
        [](bls::g1_aff& a, const uint8_t* in) { return bls::g1_decode(a, in, false); },
        [](uint8_t* o, const bls::g1_aff& a) { bls::g1_encode(o, a); });
}
int hh_bls_g2_xyzz_sum(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out) {
    return xyzz_sum<bls::fp2, bls::g2_aff>(n, pts, 96, signs, out,
        [](bls::g2_aff& a, const uint8_t* in) { return bls::g2_decode(a, in, false); },
        [](uint8_t* o, const bls::g2_aff& a) { bls::g2_encode(o, a); });
}
int hh_bn_g1_xyzz_sum(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out) {
    return xyzz_sum<bn::fp, bn::g1_aff>(n, pts, 64, signs, out,
        [](bn::g1_aff& a, const uint8_t* in) { return bn::g1_decode(a, in); },
        [](uint8_t* o, const bn::g1_aff& a) { bn::g1_encode(o, a); });
}
int hh_bn_g2_xyzz_sum(int n, const uint8_t* pts, const uint8_t* signs, uint8_t* out) {
    return xyzz_sum<bn::fp2, bn::g2_aff>(n, pts, 128, signs, out,
        [](bn::g2_aff& a, const uint8_t* in) { return bn::g2_decode(a, in); },
        [](uint8_t* o, const bn::g2_aff& a) { bn::g2_encode(o, a); });
}

// the endomorphism splits' division by z^2 (dw = 4) or |z| (dw = 2): little-endian words in and out
void hh_bls_divmod_z(int dw, const uint8_t* k32, uint8_t* q32, uint8_t* rem16) {
    uint32_t k[8], q[8];
    memcpy(k, k32, 32);
    memset(rem16, 0, 16);
    if (dw == 4) {
        uint32_t r[4];
        bls::divmod_z<4>(q, r, k);
        memcpy(rem16, r, 16);
    } else {
        uint32_t r[2];
        bls::divmod_z<2>(q, r, k);
        memcpy(rem16, r, 8);
    }
    memcpy(q32, q, 32);
}

int hh_bls_g1_fb_mul(const uint8_t* base, int nk, const uint8_t* ks, uint8_t* out) {
    return fb_mul_host<bls::fb_g1_policy, bls::fp, bls::g1_aff>(base, 0, nk, ks, out, 48, [](uint8_t* o, const bls::g1_aff& a) { bls::g1_encode(o, a); });
}
int hh_bls_g2_fb_mul(const uint8_t* base, int nk, const uint8_t* ks, uint8_t* out) {
    return fb_mul_host<bls::fb_g2_policy, bls::fp2, bls::g2_aff>(base, 0, nk, ks, out, 96, [](uint8_t* o, const bls::g2_aff& a) { bls::g2_encode(o, a); });
}
int hh_bn_g1_fb_mul(const uint8_t* base, int nk, const uint8_t* ks, uint8_t* out) {
    return fb_mul_host<bn::fb_g1_policy, bn::fp, bn::g1_aff>(base, 0, nk, ks, out, 64, [](uint8_t* o, const bn::g1_aff& a) { bn::g1_encode(o, a); });
}
int hh_bn_g2_fb_mul(const uint8_t* base, int nk, const uint8_t* ks, uint8_t* out) {
    return fb_mul_host<bn::fb_g2_policy, bn::fp2, bn::g2_aff>(base, 0, nk, ks, out, 128, [](uint8_t* o, const bn::g2_aff& a) { bn::g2_encode(o, a); });
}
// the membership verdict the finished table gives for a base that is on the curve (status of the other rules returned):
// group 1 / 2 of BLS12-381 (flags: FLAG_UNCOMPRESSED for the 96 / 192-byte form), group 2 of bn254
int hh_bls_fb_member(int grp, const uint8_t* base, int flags, int* member) {
    uint8_t sink[96];
    const uint8_t one[32] = {0};
    if (grp == 1) return fb_mul_host<bls::fb_g1_policy, bls::fp, bls::g1_aff>(base, (uint32_t)flags, 1, one, sink, 48, [](uint8_t* o, const bls::g1_aff& a) { bls::g1_encode(o, a); }, member);
    return fb_mul_host<bls::fb_g2_policy, bls::fp2, bls::g2_aff>(base, (uint32_t)flags, 1, one, sink, 96, [](uint8_t* o, const bls::g2_aff& a) { bls::g2_encode(o, a); }, member);
}
int hh_bn4_g2_fb_member(const uint8_t* base, int* member) {
    uint8_t sink[128];
    const uint8_t one[32] = {0};
    return fb_mul_host<bn4::fb_g2_policy, bn4::fp2, bn4::g2_aff>(base, 0, 1, one, sink, 128, [](uint8_t* o, const bn4::g2_aff& a) { bn4::g2_encode(o, a); }, member);
}

int hh_bls_g1_coop(const uint8_t* ops, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    return coop_run<bls::fp, bls::g1_aff>((const char*)ops, a, b, 48, out,
        [](bls::g1_aff& p, const uint8_t* in) { return bls::g1_decode(p, in, false); },
        [](uint8_t* o, const bls::g1_aff& p) { bls::g1_encode(o, p); });
}
int hh_bls_g2_coop(const uint8_t* ops, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    return coop_run<bls::fp2, bls::g2_aff>((const char*)ops, a, b, 96, out,
        [](bls::g2_aff& p, const uint8_t* in) { return bls::g2_decode(p, in, false); },
        [](uint8_t* o, const bls::g2_aff& p) { bls::g2_encode(o, p); });
}
int hh_bn_g1_coop(const uint8_t* ops, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    return coop_run<bn::fp, bn::g1_aff>((const char*)ops, a, b, 64, out,
        [](bn::g1_aff& p, const uint8_t* in) { return bn::g1_decode(p, in); },
        [](uint8_t* o, const bn::g1_aff& p) { bn::g1_encode(o, p); });
}

// jac_table8_to_affine (curve.cuh): the ladders' window table (j + 1) P, j < 8, with the entries named in `inf_mask`
// replaced by the point at infinity, normalised by the shared inversion; out = 8 encodings
int hh_bls_g1_table8(const uint8_t* pt, int inf_mask, uint8_t* out) {
    bls::g1_aff a;
    if (bls::g1_decode(a, pt, false)) return 1;
    bls::g1_jac p, tab[8];
    jac_from_aff(p, a);
    tab[0] = p;
    jac_dbl(tab[1], p);
    for (int j = 2; j < 8; j++) jac_add(tab[j], tab[j - 1], p);
    for (int j = 0; j < 8; j++)
        if ((inf_mask >> j) & 1) jac_set_inf(tab[j]);
    jac_table8_to_affine(tab);
    for (int j = 0; j < 8; j++) {
        bls::g1_aff e;
        e.x = tab[j].X;
        e.y = tab[j].Y;
        e.inf = jac_is_inf(tab[j]);
        bls::fp one;
        fp_one(one);
        if (!e.inf && !fp_eq(tab[j].Z, one)) return 2;  // finite entries must come back with Z = 1
        bls::g1_encode(out + 48 * j, e);
    }
    return 0;
}

// flag-aware variants (KYB_F_UNCOMPRESSED / _OUT / TRUSTED): ints come before the output buffers
int hh_bls_g1_mul_f(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) {
    return bls::g1_mul_wire(out, k, pt, (uint32_t)flags);
}
int hh_bls_g2_mul_f(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) {
    return bls::g2_mul_wire(out, k, pt, (uint32_t)flags);
}
int hh_bls_g1_unmarshal(const uint8_t* in, int flags, uint8_t* out) { return bls::g1_unmarshal_wire(out, in, (uint32_t)flags); }
int hh_bls_g2_unmarshal(const uint8_t* in, int flags, uint8_t* out) { return bls::g2_unmarshal_wire(out, in, (uint32_t)flags); }
int hh_bls_g1_decode_unc(const uint8_t* in, int validate) { bls::g1_aff a; return bls::g1_decode_unc(a, in, validate != 0); }
int hh_bls_g2_decode_unc(const uint8_t* in, int validate) { bls::g2_aff a; return bls::g2_decode_unc(a, in, validate != 0); }


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
int hh_bn_g2_mul_f(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) { return bn::g2_mul_wire(out, k, pt, (uint32_t)flags); }
int hh_bn_hash_g1(const uint8_t* msg, int len, uint8_t* out) { return bn::hash_g1_wire(out, msg, (size_t)len); }
void hh_bn4_fp_inv(const uint8_t* a32, uint8_t* out32) {
    bn4::fp a, r;
    bn4::fp_decode(a, a32);
    fp_inv(r, a);
    bn4::fp_encode(out32, r);
}
// bn254: the same library at alt_bn128's constants with the strict decoding rules, and its Keccak / SvdW hash
int hh_bn4_g1_decode(const uint8_t* in) { bn4::g1_aff a; return bn4::g1_decode(a, in); }
int hh_bn4_g2_decode(const uint8_t* in, int check) { bn4::g2_aff a; return bn4::g2_decode(a, in, check != 0); }
int hh_bn4_g1_mul(const uint8_t* k, const uint8_t* pt, uint8_t* out) { return bn4::g1_mul_wire(out, k, pt); }
int hh_bn4_g2_mul(const uint8_t* k, const uint8_t* pt, int flags, uint8_t* out) { return bn4::g2_mul_wire(out, k, pt, (uint32_t)flags); }
int hh_bn4_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { return bn4::g1_add_wire(out, a, b); }
int hh_bn4_g2_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { return bn4::g2_add_wire(out, a, b); }
void hh_bn4_keccak256(const uint8_t* msg, int len, uint8_t* out32) {
    Keccak256 c;
    c.init();
    c.update(msg, (size_t)len);
    uint8_t d[32];
    c.finish(d);
    memcpy(out32, d, 32);
}
void hh_bn4_map_to_point(const uint8_t* u32, uint8_t* out64) {
    bn4::fp u;
    bn4::fp_decode(u, u32);
    bn4::g1_jac r;
    bn4::map_to_point(r, u);
    bn4::g1_aff a;
    jac_to_aff(a, r);
    bn4::g1_encode(out64, a);
}
static DstArg mk_dst(const uint8_t* dst, int len) {
    DstArg d;
    memset(&d, 0, sizeof d);
    memcpy(d.b, dst, (size_t)len);
    d.len = (uint32_t)len;
    return d;
}
int hh_bn4_hash_g1(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bn4::hash_g1_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
int hh_bn_hash_g1_svdw(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bn::hash_g1_svdw_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
int hh_bls_hash_g1(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bls::hash_g1_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
int hh_bls_hash_g2(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    return bls::hash_g2_wire(out, msg, (size_t)len, mk_dst(dst, dlen));
}
// Raw-limb access to the GF(2^255 - 19) routines (ten int32 limbs, radix 2^25.5): op 0 = mul, 1 = sq, 2 = sq2,
// 3 = sq_sel(dbl = false), 4 = sq_sel(dbl = true).  The tests drive the limbs to the input bounds of fe25519.cuh.
void hh_ed_fe_op(int op, const uint8_t* f40, const uint8_t* g40, uint8_t* out40) {
    fe f, g, h;
    memcpy(f.v, f40, 40);
    memcpy(g.v, g40, 40);
    switch (op) {
        case 0: fe_mul(h, f, g); break;
        case 1: fe_sq(h, f); break;
        case 2: fe_sq2(h, f); break;
        case 3: fe_sq_sel(h, f, false); break;
        default: fe_sq_sel(h, f, true); break;
    }
    memcpy(out40, h.v, 40);
}
// ge25519.cuh ed_effective_scalar: k32 (little-endian) -> magnitude in out32, returns the sign (1 = negative)
int hh_ed_effective_scalar(const uint8_t* k32, uint8_t* out32) {
    uint32_t k[8];
    memcpy(k, k32, 32);
    const bool neg = ed_effective_scalar(k);
    memcpy(out32, k, 32);
    return neg ? 1 : 0;
}
// out = k * P with the kernels' own building blocks: decode, signed radix-16 recoding, 8-entry cached table,
// 4 doublings + 1 addition per window (the walk of ed25519_mul_kernel, ge.go:443-502), encode.  vartime = the
// all-256-bits semantics of ge_mult_vartime.go.  Returns 0, or 1 when P does not decode.
int hh_ed_mul(const uint8_t* k32, const uint8_t* p32, int vartime, uint8_t* out32) {
    uint32_t kw[8], pw[8], ow[8];
    memcpy(kw, k32, 32);
    memcpy(pw, p32, 32);
    ge_p3 P;
    if (!ge_p3_fromwords(P, pw)) {
        memset(out32, 0, 32);
        return 1;
    }
    int8_t e[65];
    recode16(e, kw, vartime != 0);
    ge_cached tab[8];
    ge_p3 cur = P;
    ge_p1p1 t;
    ge_p3_to_cached(tab[0], P);
    for (int i = 1; i < 8; i++) {
        ge_add(t, cur, tab[0]);
        ge_p1p1_to_p3(cur, t);
        ge_p3_to_cached(tab[i], cur);
    }
    ge_p3 acc;
    ge_p3_0(acc);
    for (int i = 64; i >= 0; i--) {
        if (i != 64) {
            ge_p2 d;
            ge_dbl(t, acc.X, acc.Y, acc.Z);
            for (int r = 0; r < 3; r++) {
                ge_p1p1_to_p2(d, t);
                ge_dbl(t, d.X, d.Y, d.Z);
            }
            ge_p1p1_to_p3(acc, t);
        }
        const int dgt = e[i];
        if (dgt != 0) {
            ge_cached c = tab[(dgt < 0 ? -dgt : dgt) - 1];
            ge_cached_cneg(c, dgt < 0);
            ge_add(t, acc, c);
            ge_p1p1_to_p3(acc, t);
        }
    }
    ge_p3_towords(ow, acc);
    memcpy(out32, ow, 32);
    return 0;
}
// out = k * P by plain double-and-add over the MIXED addition: P as a precomputed (y + x, y - x, 2dxy) entry built
// the way the fixed-base table and the MSM decode build theirs (sums reduced by a multiplication by one), ge_madd +
// ge_p1p1_to_p3 per set bit, ge_dbl per bit.  Covers the formulas of ed25519_mul_base_kernel and the MSM buckets.
int hh_ed_mul_madd(const uint8_t* k32, const uint8_t* p32, uint8_t* out32) {
    uint32_t pw[8], ow[8];
    memcpy(pw, p32, 32);
    ge_p3 P;
    if (!ge_p3_fromwords(P, pw)) {
        memset(out32, 0, 32);
        return 1;
    }
    fe zi, x, y, one;
    fe_invert(zi, P.Z);
    fe_mul(x, P.X, zi);
    fe_mul(y, P.Y, zi);
    fe_1(one);
    ge_precomp pre;
    fe_add(pre.ypx, y, x);
    fe_sub(pre.ymx, y, x);
    fe_mul(pre.ypx, pre.ypx, one);
    fe_mul(pre.ymx, pre.ymx, one);
    fe_mul(pre.xy2d, x, y);
    fe_mul(pre.xy2d, pre.xy2d, fe_d2());
    ge_p3 acc;
    ge_p3_0(acc);
    ge_p1p1 t;
    for (int bit = 255; bit >= 0; bit--) {
        ge_dbl(t, acc.X, acc.Y, acc.Z);
        ge_p1p1_to_p3(acc, t);
        if ((k32[bit >> 3] >> (bit & 7)) & 1) {
            ge_precomp q = pre;
            ge_precomp_cneg(q, false);
            ge_madd(t, acc, q);
            ge_p1p1_to_p3(acc, t);
        }
    }
    ge_p3_towords(ow, acc);
    memcpy(out32, ow, 32);
    return 0;
}
#if defined(KYB_FE_AUDIT)
// bound audit build (tests/_host_harness.py lib_audit): the largest mag_f * mag_g met by a multiplication, scaled by
// 10^6 (the call interface returns ints), and a reset
// lazy-limb audit (fp_limbs.cuh KYB_LZ_AUDIT): failures so far / the largest product bound met, as a fraction (in
// 1e-6) of the field's R / p it must stay below -- the reset returns the failures and clears both
int hh_lz_audit_failures() {
#ifdef KYB_LZ_AUDIT
    return lz_audit_failures();
#else
    return -1;
#endif
}
int hh_lz_audit_max_product() {
#ifdef KYB_LZ_AUDIT
    return (int)lz_audit_max_product();
#else
    return -1;
#endif
}
int hh_lz_audit_reset() {
#ifdef KYB_LZ_AUDIT
    const int f = lz_audit_failures();
    lz_audit_failures() = 0;
    lz_audit_max_product() = 0;
    return f;
#else
    return -1;
#endif
}
int hh_fe_audit_max_micro() { return (int)(fe_audit_max() * 1e6); }
int hh_fe_audit_max19_micro() { return (int)(fe_audit_max19() * 1e6); }
int hh_fe_audit_reset() {
    fe_audit_max() = 0;
    fe_audit_max19() = 0;
    return 0;
}
#endif
void hh_ed_hash(const uint8_t* msg, int len, const uint8_t* dst, int dlen, uint8_t* out) {
    EdDstArg d;
    memset(&d, 0, sizeof d);
    memcpy(d.b, dst, (size_t)dlen);
    d.len = (uint32_t)dlen;
    ed_hash_wire(out, msg, (size_t)len, d);
}
// ---- scalar-field Horner (scalar_field.cuh): suite 0 = Ed25519 (little-endian), 1 = BLS12-381, 2 = bn256, 3 = bn254
int hh_scalar_poly_eval(int suite, int n, const uint8_t* idx4, int t, const uint8_t* coeffs, uint8_t* out) {
    using namespace kyb::sf;
    const Mod m = suite == 0 ? make_mod(Q_ED25519, false) : suite == 1 ? make_mod(Q_BLS12381, true) : suite == 2 ? make_mod(Q_BN256, true) : make_mod(Q_BN254, true);
    std::vector<uint32_t> cm((size_t)8 * (t ? t : 1));
    for (int j = 0; j < t; j++) {
        uint32_t r[8];
        to_mont(r, coeffs + 32 * j, m);
        for (int i = 0; i < 8; i++) cm[8 * (size_t)j + i] = r[i];
    }
    for (int i = 0; i < n; i++) {
        uint32_t ix;
        memcpy(&ix, idx4 + 4 * i, 4);
        horner(out + 32 * i, ix, (size_t)t, cm.data(), m);
    }
    return 0;
}
// ---- the Miller lines of a fixed G2 point (bls12381_keylines.cuh): 68 x 4 x 48 bytes, each the little-endian integer
// c 2^392 mod p the tower machine's same-key verification program takes as a constant
int hh_bls_g2_key_lines(const uint8_t* q96, uint8_t* out) {
    bls::g2_aff q;
    const int st = bls::g2_decode(q, q96, true);
    if (st || q.inf) return st ? st : 64;
    static uint32_t lines[bls::KEYLINE_STEPS][4][12];
    if (!bls::g2_key_lines(lines, q)) return 65;
    memcpy(out, lines, sizeof lines);
    return 0;
}
// the same walk on the limb-per-lane arithmetic (g2_key_lines_rows, emulated lane by lane): must give the same words
int hh_bls_g2_key_lines_rows(const uint8_t* q96, uint8_t* out, int* overflows) {
    bls::g2_aff q;
    const int st = bls::g2_decode(q, q96, true);
    if (st || q.inf) return st ? st : 64;
    static uint32_t lines[bls::KEYLINE_STEPS][4][12];
    static bls::KeyLinesMem mem;
    rowfp::overflow_count() = 0;
    const bool ok = bls::g2_key_lines_rows(mem, lines, q.x.c0.v, q.x.c1.v, q.y.c0.v, q.y.c1.v);
    *overflows = rowfp::overflow_count();
    if (!ok) return 65;
    memcpy(out, lines, sizeof lines);
    return 0;
}
// the walk with the r-torsion rule read off its end (member_test): an uncompressed key, curve equation checked, subgroup not;
// 0 = accepted, 65 = rejected; tw = the walk's end T = |z| q (six packed Montgomery residues, valid up to the rejection)
int hh_bls_g2_key_walk_member(const uint8_t* q192, uint8_t* tw, int* overflows) {
    bls::g2_aff q;
    const int st = bls::g2_decode_unc(q, q192, true, false);
    if (st || q.inf) return st ? st : 64;
    static uint32_t lines[bls::KEYLINE_STEPS][4][12];
    static bls::KeyLinesMem mem;
    rowfp::overflow_count() = 0;
    const bool ok = bls::g2_key_lines_rows(mem, lines, q.x.c0.v, q.x.c1.v, q.y.c0.v, q.y.c1.v, true);
    *overflows = rowfp::overflow_count();
    for (int j = 0; j < 6; j++) {  // out of the Montgomery domain: plain little-endian integers
        bls::fp f;
        for (int w = 0; w < 12; w++) f.v[w] = mem.tw[j][w];
        uint32_t words[12];
        kyb::fp_to_words<bls::FC>(words, f);
        memcpy(tw + 48 * j, words, 48);
    }
    return ok ? 0 : 65;
}
// a compressed key's UnmarshalBinary without the r-torsion rule, the square root's two powers on the rows (g2_decode_rows);
// returns the status; out = x.c0, x.c1, y.c0, y.c1 as plain little-endian integers (when accepted and finite)
int hh_bls_g2_decode_rows(const uint8_t* in96, uint8_t* out, int* inf, int* overflows) {
    static bls::KeyDecodeMem mem;
    static uint32_t qw[4][12];
    rowfp::overflow_count() = 0;
    memset(qw, 0, sizeof qw);
    const int st = bls::g2_decode_rows(mem, qw, inf, in96);
    *overflows = rowfp::overflow_count();
    for (int j = 0; j < 4; j++) {
        bls::fp f;
        for (int w = 0; w < 12; w++) f.v[w] = qw[j][w];
        uint32_t words[12];
        kyb::fp_to_words<bls::FC>(words, f);
        memcpy(out + 48 * j, words, 48);
    }
    return st;
}
// ---- G1Elt.Mul on four cooperating lanes (bls12381_g1coop.cuh) with four THREADS as the lanes of one group: the whole
// ladder -- table, 34 windows, the z^2 half from the beta x slots -- against the per-lane routine's answer
int hh_bls_g1_mul_coop(const uint8_t* k32, const uint8_t* pt, int flags, uint8_t* out) {
    using namespace kyb::bls;
    g1_aff a;
    const int st = g1_decode_f(a, pt, (uint32_t)flags, 0);
    if (st) return st;
    if (a.inf) {
        g1_encode_f(out, a, (uint32_t)flags);
        return 0;
    }
    uint32_t k[8];
    scalar_from_be(k, k32);
    static int8_t e0[g1coop::NDIG], e1[g1coop::NDIG];
    g1coop::digits(e0, e1, k);
    static g1coop::Slot S[g1coop::NS];
    static uint32_t fl[2];
    S[g1coop::TAB].f = a.x;
    S[g1coop::TAB + 1].f = a.y;
    fp_one(S[g1coop::TAB + 2].f);
    pthread_barrier_init(&g_coop_barrier, nullptr, 4);
    std::thread th[4];
    for (int r = 0; r < 4; r++) th[r] = std::thread([=]() { g1coop::ladder(S, fl, r, e0, e1); });
    for (int r = 0; r < 4; r++) th[r].join();
    pthread_barrier_destroy(&g_coop_barrier);
    g1_jac p;
    p.X = S[g1coop::ACC].f;
    p.Y = S[g1coop::ACC + 1].f;
    p.Z = S[g1coop::ACC + 2].f;
    jac_to_aff(a, p);
    g1_encode_f(out, a, (uint32_t)flags);
    return 0;
}
// the subgroup rule on four cooperating lanes (g1coop::member), threads as lanes; *verdict = 1 for a member
int hh_bls_g1_member_coop(const uint8_t* pt, int* verdict) {
    using namespace kyb::bls;
    g1_aff a;
    const int st = g1_decode(a, pt, false);
    if (st || a.inf) return st ? st : 64;
    static g1coop::Slot S[g1coop::NS];
    static uint32_t fl[2];
    static int res[4];
    S[g1coop::TAB].f = a.x;
    S[g1coop::TAB + 1].f = a.y;
    fp_one(S[g1coop::TAB + 2].f);
    pthread_barrier_init(&g_coop_barrier, nullptr, 4);
    std::thread th[4];
    for (int r = 0; r < 4; r++) th[r] = std::thread([=]() { res[r] = g1coop::member(S, fl, r) ? 1 : 0; });
    for (int r = 0; r < 4; r++) th[r].join();
    pthread_barrier_destroy(&g_coop_barrier);
    *verdict = res[0];
    return 0;
}
// G2Elt.Mul on four cooperating lanes (g2coop), threads as lanes; member != null: only the subgroup rule's verdict
int hh_bls_g2_mul_coop(const uint8_t* k32, const uint8_t* pt, int flags, uint8_t* out, int* member) {
    using namespace kyb::bls;
    g2_aff a;
    const int st = member ? g2_decode(a, pt, false) : g2_decode_f(a, pt, (uint32_t)flags, 0);
    if (st) return st;
    if (a.inf) {
        if (member) return 64;
        g2_encode_f(out, a, (uint32_t)flags);
        return 0;
    }
    uint32_t k[8];
    scalar_from_be(k, k32);
    static int8_t e[g2coop::NH][g2coop::NDIG];
    g2coop::digits(e, k);
    static g2coop::Slot S[g2coop::NS], C[g2coop::NCONST];
    static uint32_t fl[2];
    static int res[4];
    g2coop::constants(C);
    S[g2coop::TAB].f = a.x;
    S[g2coop::TAB + 1].f = a.y;
    fp2_one(S[g2coop::TAB + 2].f);
    pthread_barrier_init(&g_coop_barrier, nullptr, 4);
    std::thread th[4];
    const bool only_member = member != nullptr;
    for (int r = 0; r < 4; r++)
        th[r] = std::thread([=]() {
            if (only_member) res[r] = g2coop::member(S, C, fl, r) ? 1 : 0;
            else g2coop::ladder(S, C, fl, r, &e[0][0]);
        });
    for (int r = 0; r < 4; r++) th[r].join();
    pthread_barrier_destroy(&g_coop_barrier);
    if (member) {
        *member = res[0];
        return 0;
    }
    g2_jac p;
    p.X = S[g2coop::ACC].f;
    p.Y = S[g2coop::ACC + 1].f;
    p.Z = S[g2coop::ACC + 2].f;
    jac_to_aff(a, p);
    g2_encode_f(out, a, (uint32_t)flags);
    return 0;
}
}

// ---- rowfp.cuh: the limb-per-lane field arithmetic, emulated lane by lane (V32 = 64 lanes).  Operands and results travel
// as raw limbs, 4 rows x 16 lanes of uint32 (lanes 13..15 zero), so the tests see the redundant representation itself.
static rowfp::V32 row_in(const uint32_t* x) { rowfp::V32 v; for (int i = 0; i < 64; i++) v.v[i] = x[i]; return v; }
static void row_out(uint32_t* o, const rowfp::V32& v) { for (int i = 0; i < 64; i++) o[i] = v.v[i]; }
// op 0: mul, 1: add2, 2: dbl, 3: triple, 4: a - b + 3p, 5: a - b + 5p, 6: a - b + 8p.  Returns the number of 64-bit
// accumulator overflows the emulation saw (must be 0).
extern "C" {
int hh_row_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    using C = bls::FC;
    rowfp::overflow_count() = 0;
    const auto cx = rowfp::make_ctx<C>();
    const auto dc = rowfp::make_dbl_consts<C>();
    const rowfp::V32 A = row_in(a), B = row_in(b);
    rowfp::V32 R;
    switch (op) {
        case 0: R = rowfp::mul<C>(cx, A, B); break;
        case 1: R = rowfp::add2<C>(A, B); break;
        case 2: R = rowfp::dbl<C>(A); break;
        case 3: R = rowfp::triple<C>(A); break;
        case 4: R = rowfp::sub_k<3, C>(A, B, dc.b3); break;
        case 5: R = rowfp::sub_k<5, C>(A, B, dc.b5); break;
        default: R = rowfp::sub_k<8, C>(A, B, dc.b8); break;
    }
    row_out(out, R);
    return rowfp::overflow_count();
}
// n doublings of the points (X, Y, Z) -- wave = 0: each row its own point; wave = 1: the four rows hold the same point and
// share the products (jac_dbl_wave).  maxlimb receives the largest limb seen at the end of any doubling.
int hh_row_dbl_chain(int wave, int n, const uint32_t* X, const uint32_t* Y, const uint32_t* Z, uint32_t* oX, uint32_t* oY, uint32_t* oZ,
                     uint32_t* maxlimb) {
    using C = bls::FC;
    rowfp::overflow_count() = 0;
    const auto cx = rowfp::make_ctx<C>();
    const auto dc = rowfp::make_dbl_consts<C>();
    const rowfp::V32 row = rowfp::row_of_lane();
    rowfp::JacRow<C> p{row_in(X), row_in(Y), row_in(Z)};
    uint32_t mx = 0;
    for (int k = 0; k < n; k++) {
        if (wave) rowfp::jac_dbl_wave<C>(cx, dc, row, p);
        else rowfp::jac_dbl<C>(cx, dc, p);
        for (int i = 0; i < 64; i++) {
            if (p.X.v[i] > mx) mx = p.X.v[i];
            if (p.Y.v[i] > mx) mx = p.Y.v[i];
            if (p.Z.v[i] > mx) mx = p.Z.v[i];
        }
    }
    row_out(oX, p.X);
    row_out(oY, p.Y);
    row_out(oZ, p.Z);
    *maxlimb = mx;
    return rowfp::overflow_count();
}
// the way out of the row form: a product with R mod p (value below 2p), the limbs of row 0 rippled by one lane, fp_finish;
// out = the 12 packed words of the residue, fully reduced.  Also checks load_packed against the limbs it came from.
int hh_row_finish(const uint32_t* x, uint32_t* out12) {
    using C = bls::FC;
    rowfp::overflow_count() = 0;
    const auto cx = rowfp::make_ctx<C>();
    const rowfp::V32 y = rowfp::below_2p<C>(cx, row_in(x));
    Fp<C> f;
    rowfp::finish_limbs<C>(f, y.v);
    for (int j = 0; j < C::NWORDS; j++) out12[j] = f.v[j];
    const rowfp::V32 back = rowfp::load_packed<C>(f.v);
    Fp<C> g;
    rowfp::finish_limbs<C>(g, back.v + 32);  // row 2 of the reloaded element: the same limbs in every row
    for (int j = 0; j < C::NWORDS; j++)
        if (g.v[j] != f.v[j]) return -1;
    return rowfp::overflow_count();
}
}
