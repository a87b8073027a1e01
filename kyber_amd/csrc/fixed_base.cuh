// Fixed-base scalar multiplication for the same-base batches of the pairing suites: out_i = k_i * P for ONE point P and
// many scalars -- share.PriPoly.Commit (share/poly.go:143-149: every coefficient times the same base), key generation
// (x * G2.Base()), `Point.Mul(s, nil)`.  The reference runs its variable-base ladder n times; with the base shared, a
// table of its multiples turns a multiplication into ~26 table additions and NO doublings:
//
//   a sub-scalar s = sum_w d_w 1024^w with signed digits d_w in [-511, 512], and T[w][e] = (e + 1) 1024^w P in affine
//   form (512 entries per window), so  s P = sum_w sign(d_w) T[w][|d_w| - 1]:  mixed additions in XYZZ form (curve.cuh,
//   8M + 2S each) + one inversion, about a sixth of the GLV ladder's field multiplications and a tenth of the plain
//   ladder's (bn256 G2).
//
// How many windows, and of what, is the group's POLICY (round 4):
//   * plain (bn256 / bn254, both groups): ONE sub-scalar -- k itself, 26 windows; no endomorphism is involved, so the
//     result is the integer multiple for EVERY decodable base (bn256's G2 points outside the order-n subgroup included)
//     and every 256-bit scalar, like the reference's double-and-add;
//   * BLS12-381 G1: k = q z^2 + rem (plain long division), two sub-scalars of 13 windows over the entry's two IMAGES
//     (x, y) and z^2 P = (beta x, -y) -- 26 additions as before, but the table's doubling chain, the one serial part of
//     a table build, is 120 doublings instead of 250;
//   * BLS12-381 G2: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3, four sub-scalars of 7 windows over the images
//     |z|^j Q = (-1)^j psi^j(Q): 28 additions, a chain of 60 doublings instead of 250.
//   The splits are integer identities, the images hold on the order-r subgroup -- which is what UnmarshalBinary
//   guarantees for BLS12-381 (and KYB_F_TRUSTED vouches for).
//
// Membership of the base, where it is a rule of UnmarshalBinary and the caller has not vouched for the point, is not
// tested in the lone lane that decodes the base (a 128-bit ladder there cost as much as half the doubling chain): the
// finished table's plain image answers it -- Scott's criterion z^2 P = -phi(P) (G1: 13 additions), |z| Q = -psi(Q) (G2:
// 7), [n] Q = infinity (bn254 G2: 26) -- in one lane between the table and the batch (0.1 - 0.3 ms, and only when the
// table is new); a base that fails gets status 2 for every coefficient, as the reference's UnmarshalBinary would fail
// once, for all of them.
//
// The table costs the chain (four cooperating lanes, coop_slots.cuh) + NW x 512 independent small multiples: it pays
// for itself from ~2^17 scalars, or at any batch size once it exists -- the workspace keeps the last base's table per
// (suite, group, stream), and the chain kernel recognises the base on the device (no host round trip).  Table: 1.3 MB
// (BLS12-381 G1) .. 2.8 MB (G2): inside the 4 MB of L2 each XCD has.
#pragma once
#include "coop_slots.cuh"
#include "curve.cuh"
#if defined(__HIPCC__)
#include "context.h"
#endif

namespace kyb {
namespace fb {

constexpr int WBITS = 10, NENT = 1 << (WBITS - 1), WIRE_MAX = 192;
constexpr uint64_t MAGIC = 0x6b79626662763033ull;  // "kybfbv03"

// head of the workspace: which base the table below belongs to
struct Header {
    uint64_t magic;
    uint32_t key_flags;   // the decode-relevant flag bits the base was decoded under
    uint32_t key_len;     // bytes of its wire form
    uint32_t status;      // UnmarshalBinary's verdict on it (ST_OK = 0); the membership verdict arrives late (below)
    uint32_t inf;         // it is the point at infinity
    uint32_t fresh;       // set by the chain kernel of THIS call: the table has to be rebuilt
    uint32_t member_pending;  // the base's subgroup membership is still to be read off the finished table
    uint8_t key[WIRE_MAX];
};

// affine multiple under the policy's NI images; (0, 0) -- not a point of y^2 = x^3 + b, b != 0 -- stands for infinity
// (only a plain table of a base outside the prime-order subgroup can hold one)
template <class F, int NI>
struct Entry {
    F x[NI], y[NI];
};

// The plain policy: one sub-scalar, the 256-bit integer itself
template <class F>
struct Plain {
    static constexpr int NI = 1, NW = (256 + WBITS) / WBITS;
    static_assert(NW * WBITS >= 257, "the last window must hold the recoding's carry");
    KYB_HD static void split(uint32_t (&sub)[1][8], const uint32_t (&k)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) sub[0][i] = k[i];
    }
    KYB_HD static void images(Entry<F, 1>& e, const F& x, const F& y) {
        e.x[0] = x;
        e.y[0] = y;
    }
};

// the next signed radix-2^WBITS digit of s, in [-(NENT - 1), NENT]; s is shifted down by one window
KYB_HD int next_digit(uint32_t (&s)[8], int& carry) {
    const int v = (int)(s[0] & ((1u << WBITS) - 1)) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) s[i] = (s[i] >> WBITS) | (s[i + 1] << (32 - WBITS));
    s[7] >>= WBITS;
    carry = v > NENT ? 1 : 0;
    return v - (carry << WBITS);
}

// q[w] = 2^(WBITS w) P  (Jacobian): the one serial part of the table (one lane; the kernel below runs it on four)
template <int NW, class F>
KYB_HD void chain(Jac<F> (&q)[NW], const Aff<F>& base) {
    jac_from_aff(q[0], base);
#pragma unroll 1
    for (int w = 1; w < NW; w++) {
        Jac<F> t = q[w - 1];
#pragma unroll 1
        for (int i = 0; i < WBITS; i++) jac_dbl_inl(t, t);
        q[w] = t;
    }
}

// a = (j + 1) Q in affine form, j in [0, NENT)
template <class F>
KYB_HD void entry(Aff<F>& a, const Jac<F>& q, int j) {
    const int m = j + 1;  // 1 .. NENT
    Jac<F> acc;
    jac_set_inf(acc);
#pragma unroll 1
    for (int bit = WBITS - 1; bit >= 0; bit--) {
        jac_dbl_inl(acc, acc);
        if ((m >> bit) & 1) jac_add_inl<F, false>(acc, acc, q);
    }
    jac_to_aff(a, acc);
}
template <class P, class F>
KYB_HD void entry_images(Entry<F, P::NI>& e, const Jac<F>& q, int j) {
    Aff<F> a;
    entry(a, q, j);
    if (a.inf) {
#pragma unroll
        for (int i = 0; i < P::NI; i++) {
            f_zero(e.x[i]);
            f_zero(e.y[i]);
        }
    } else {
        P::images(e, a.x, a.y);
    }
}

// acc += s * (image j of the table's base), windows 0 .. nw - 1 of s (consumed).  The accumulator is the field's run
// form (curve.cuh XyzzSel: limbs for BLS12-381 G1, where the digit's sign is applied inside the addition).
template <int NI, class F>
KYB_HD void walk(typename XyzzSel<F>::type& acc, uint32_t (&s)[8], int j, int nw, const Entry<F, NI>* __restrict__ tab) {
    int carry = 0;
#pragma unroll 1
    for (int w = 0; w < nw; w++) {
        const int dw = next_digit(s, carry);
        if (dw == 0) continue;
        const int a = dw < 0 ? -dw : dw;
        const F* e = reinterpret_cast<const F*>(tab + (w * NENT + (a - 1)));  // x[0 .. NI), y[0 .. NI)
        const F x = e[j];
        const F y = e[NI + j];
        if (f_is_zero(x) & f_is_zero(y)) continue;  // that multiple of the base is the point at infinity
        XyzzSel<F>::madd(acc, x, y, dw < 0);
    }
}
// r = k * P from P's table under policy P_; r comes back in Jacobian form
template <class P_, class F>
KYB_HD void mul(Jac<F>& r, const uint32_t (&k)[8], const Entry<F, P_::NI>* __restrict__ tab) {
    uint32_t sub[P_::NI][8];
    P_::split(sub, k);
    typename XyzzSel<F>::type acc;
    XyzzSel<F>::identity(acc);
#pragma unroll 1
    for (int j = 0; j < P_::NI; j++) {
        uint32_t s[8];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = sub[j][i];
        walk<P_::NI>(acc, s, j, P_::NW, tab);
    }
    XyzzSel<F>::finish(r, acc);
}
// r = s * P for an integer s below 2^(WBITS nw - 1), from the PLAIN image alone: the integer multiple for any point of
// the curve -- what the membership criteria are evaluated with
template <int NI, class F>
KYB_HD void mul_plain(Jac<F>& r, const uint32_t (&k)[8], int nw, const Entry<F, NI>* __restrict__ tab) {
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = k[i];
    typename XyzzSel<F>::type acc;
    XyzzSel<F>::identity(acc);
    walk<NI>(acc, s, 0, nw, tab);
    XyzzSel<F>::finish(r, acc);
}


#if defined(__HIPCC__)
// ---- kernels and the enqueue routine, over a per-(suite, group) traits type T:
//   F; P (the policy: NI, NW, split, images); decode_on_curve(Aff<F>&, wire, flags) -> status (every rule of
//   UnmarshalBinary except subgroup membership); needs_member(flags); member(Aff<F> base, table) -> bool;
//   encode(out, Aff<F>, flags); wire_size(flags); out_size(flags); scalar(k[8], wire32); KIND (workspace kind);
//   KEY_FLAGS (the flag bits decoding looks at)
constexpr size_t HDR_BYTES = 512;
template <class T>
constexpr size_t chain_bytes() { return ((sizeof(Jac<typename T::F>) * T::P::NW + sizeof(Aff<typename T::F>) + 255) / 256) * 256; }
template <class T>
constexpr size_t ws_bytes() { return HDR_BYTES + chain_bytes<T>() + sizeof(Entry<typename T::F, T::P::NI>) * T::P::NW * NENT; }
constexpr size_t PARK_MIN = size_t(1) << 14;  // batches from which the results are parked for the shared inversion
template <class T>
__device__ __forceinline__ Aff<typename T::F>* base_slot(uint8_t* ws) {  // the decoded base, kept for the membership test
    return reinterpret_cast<Aff<typename T::F>*>(ws + HDR_BYTES + sizeof(Jac<typename T::F>) * T::P::NW);
}

// Run by ONE lane: is the table already this base's?  Otherwise decode the base (every rule of UnmarshalBinary but the
// subgroup), write the header and, when there is a table to build, the chain's first point q[0] (returned in `start`).
// Returns 1 when the doubling chain has to be walked.
// chain_begin in its three parts (chain_rows_kernel puts the wave between the second and the third): is the table this
// base's already; the base's decoding; the header and the chain's first point
template <class T>
__device__ bool chain_same(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags) {
    Header* h = reinterpret_cast<Header*>(ws);
    const uint32_t kf = flags & T::KEY_FLAGS, len = (uint32_t)T::wire_size(flags);
    bool same = h->magic == MAGIC && h->key_flags == kf && h->key_len == len;
    for (uint32_t i = 0; i < len && same; i++) same = h->key[i] == base[i];
    if (same) {
        h->fresh = 0;
        h->member_pending = 0;
        return true;
    }
    h->magic = 0;
    return false;
}
template <class T>
__device__ int chain_commit(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags, int st, const Aff<typename T::F>& a,
                            Jac<typename T::F>& start);
template <class T>
__device__ int chain_begin(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags, Jac<typename T::F>& start) {
    if (chain_same<T>(ws, base, flags)) return 0;
    Aff<typename T::F> a;
    const int st = T::decode_on_curve(a, base, flags);
    return chain_commit<T>(ws, base, flags, st, a, start);
}
template <class T>
__device__ int chain_commit(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags, int st, const Aff<typename T::F>& a,
                            Jac<typename T::F>& start) {
    using F = typename T::F;
    Header* h = reinterpret_cast<Header*>(ws);
    Jac<F>* q = reinterpret_cast<Jac<F>*>(ws + HDR_BYTES);
    const uint32_t kf = flags & T::KEY_FLAGS, len = (uint32_t)T::wire_size(flags);
    const bool build = st == 0 && !a.inf;
    h->status = (uint32_t)st;
    h->inf = (st == 0 && a.inf) ? 1u : 0u;
    h->key_flags = kf;
    h->key_len = len;
    for (uint32_t i = 0; i < len; i++) h->key[i] = base[i];
    h->fresh = 1;
    h->member_pending = (build && T::needs_member(flags)) ? 1u : 0u;
    if (build) {
        jac_from_aff(start, a);
        q[0] = start;
        *base_slot<T>(ws) = a;
        return 1;
    }
    __threadfence();
    h->magic = MAGIC;
    return 0;
}
// FOUR cooperating lanes walk the doubling chain (coop_slots.cuh: a doubling is three product levels deep instead of
// seven multiplications).  One workgroup of four lanes.
template <class T>
__global__ __launch_bounds__(64, 2) void chain_kernel(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags) {
    using F = typename T::F;
    constexpr int P = 0, TMP = 3;
    __shared__ coop::Slot<F> S[TMP + coop::TEMPS];
    __shared__ int go;
    const int r = (int)threadIdx.x & 3;
    Header* h = reinterpret_cast<Header*>(ws);
    Jac<F>* q = reinterpret_cast<Jac<F>*>(ws + HDR_BYTES);
    if (threadIdx.x == 0) {
        Jac<F> t;
        go = chain_begin<T>(ws, base, flags, t);
        if (go) {
            S[P].f = t.X;
            S[P + 1].f = t.Y;
            S[P + 2].f = t.Z;
        }
    }
    __syncthreads();
    if (!go) return;  // uniform: the table is current, or there is nothing to build
#pragma unroll 1
    for (int w = 1; w < T::P::NW; w++) {
#pragma unroll 1
        for (int i = 0; i < WBITS; i++) coop::dbl<F>(S, r, P, TMP, true);
        if (r < 3) reinterpret_cast<F*>(q + w)[r] = S[P + r].f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        h->magic = MAGIC;
    }
}
// The same chain with ONE LIMB PER LANE (rowfp.cuh; traits with ROW_CHAIN: a base field of 13 x 30-bit limbs) -- round 6:
// one wave, its four rows holding the running point and sharing the seven products of a doubling.  The four-lane
// kernel above spends 1.4 ms on the 120 doublings of a BLS12-381 G1 table -- most of what a new base costs.
template <class T, class = void>
struct HasRowChain {
    static constexpr bool value = false;
};
template <class T>
struct HasRowChain<T, decltype((void)T::ROW_CHAIN)> {
    static constexpr bool value = T::ROW_CHAIN != 0;
};
template <class P, class = void>
struct HasRowSqrt {
    static constexpr bool value = false;
};
template <class P>
struct HasRowSqrt<P, decltype((void)P::ROW_SQRT)> {
    static constexpr bool value = P::ROW_SQRT != 0;
};
#if defined(KYB_ROWFP_INCLUDED)
template <class T>
__global__ __launch_bounds__(64) void chain_rows_kernel(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags) {
    using F = typename T::F;
    using C = typename T::RowC;
    using namespace rowfp;
    __shared__ uint32_t limbs[3][ROW];
    __shared__ int go;
    Header* h = reinterpret_cast<Header*>(ws);
    Jac<F>* q = reinterpret_cast<Jac<F>*>(ws + HDR_BYTES);
    const auto cx = make_ctx<C>();
    if constexpr (HasRowSqrt<typename T::P>::value) {
        // a new COMPRESSED base: its square root's power is the wave's (rowfp::pow_words), the rest of UnmarshalBinary lane 0's
        __shared__ uint32_t ptab[15][ROW], pio[C::NWORDS];
        __shared__ F sx, srhs;
        __shared__ int root, sflag;
        if (threadIdx.x == 0) {
            Jac<F> t;
            go = 0;
            root = 0;
            if (!chain_same<T>(ws, base, flags)) {
                Aff<F> a;
                if (flags & FLAG_UNCOMPRESSED) {
                    const int st = T::decode_on_curve(a, base, flags);
                    go = chain_commit<T>(ws, base, flags, st, a, t);
                } else {
                    F x, rhs;
                    int inf = 1, sf = 0;
                    const int st = T::P::decode_head(x, rhs, sf, inf, base);
                    if (st == 0 && !inf) {
                        sx = x;
                        srhs = rhs;
                        sflag = sf;
                        for (int k = 0; k < C::NWORDS; k++) pio[k] = rhs.v[k];
                        root = 1;
                    } else {
                        a.x = F{};
                        a.y = F{};
                        a.inf = true;
                        go = chain_commit<T>(ws, base, flags, st, a, t);
                    }
                }
            }
            __threadfence_block();
        }
        __syncthreads();
        if (root) {  // (uniform)
            const V32 y = below_2p<C>(cx, pow_words<C>(cx, load_packed<C>(pio), ptab, C::SQRT_EXP, C::SQRT_BITS));
            store_row(limbs[0], y);
            __syncthreads();
            if (threadIdx.x == 0) {
                Jac<F> t;
                F y0;
                Aff<F> a;
                finish_limbs<C>(y0, limbs[0]);
                const int st = T::P::decode_tail(a, sx, srhs, y0, sflag);
                go = chain_commit<T>(ws, base, flags, st, a, t);
                __threadfence_block();
            }
            __syncthreads();
        }
    } else {
        if (threadIdx.x == 0) {
            Jac<F> t;
            go = chain_begin<T>(ws, base, flags, t);
            __threadfence_block();
        }
        __syncthreads();
    }
    if (!go) return;
    const auto dc = make_dbl_consts<C>();
    const V32 row = row_of_lane();
    JacRow<C> pt{load_packed<C>(q[0].X.v), load_packed<C>(q[0].Y.v), load_packed<C>(q[0].Z.v)};
#pragma unroll 1
    for (int w = 1; w < T::P::NW; w++) {
#pragma unroll 1
        for (int i = 0; i < WBITS; i++) jac_dbl_wave<C>(cx, dc, row, pt);
        const V32 X = below_2p<C>(cx, pt.X), Y = below_2p<C>(cx, pt.Y), Z = below_2p<C>(cx, pt.Z);
        if (threadIdx.x < ROW) {
            limbs[0][threadIdx.x] = X;
            limbs[1][threadIdx.x] = Y;
            limbs[2][threadIdx.x] = Z;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            F f;
            finish_limbs<C>(f, limbs[threadIdx.x]);
            reinterpret_cast<F*>(q + w)[threadIdx.x] = f;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        h->magic = MAGIC;
    }
}
#endif
// One lane per table entry
template <class T>
__global__ __launch_bounds__(64, KYB_TU_WAVES) void table_kernel(uint8_t* __restrict__ ws) {
    using F = typename T::F;
    const Header* h = reinterpret_cast<const Header*>(ws);
    if (!h->fresh || h->status || h->inf) return;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= T::P::NW * NENT) return;
    const int w = t / NENT, j = t - w * NENT;
    const Jac<F> q = reinterpret_cast<const Jac<F>*>(ws + HDR_BYTES)[w];
    Entry<F, T::P::NI> e;
    entry_images<typename T::P>(e, q, j);
    reinterpret_cast<Entry<F, T::P::NI>*>(ws + HDR_BYTES + chain_bytes<T>())[t] = e;
}
// One lane, between the table and the batch: a fresh base's subgroup membership, read off the table's plain image (a
// kernel of its own so that the multiplication kernel's register allocation is the walk's alone)
template <class T>
__global__ __launch_bounds__(64, T::MUL_WAVES) void member_kernel(uint8_t* __restrict__ ws) {
    using F = typename T::F;
    Header* h = reinterpret_cast<Header*>(ws);
    if (threadIdx.x != 0 || blockIdx.x != 0 || !h->member_pending) return;
    const auto* tab = reinterpret_cast<const Entry<F, T::P::NI>*>(ws + HDR_BYTES + chain_bytes<T>());
    const Aff<F> a = *base_slot<T>(ws);
    if (!T::member(a, tab)) h->status = 2u;  // ST_NOT_IN_SUBGROUP: every coefficient fails alike
    h->member_pending = 0;
}
// One lane per scalar.  `park` (large batches): the Jacobian results go there and encode_kernel turns them into wire
// bytes eight at a time with ONE inversion (Montgomery's trick) -- the division-step inversion is ~22 k instructions,
// an eighth of the walk's 26 additions, per element otherwise.
template <class T>
__global__ __launch_bounds__(64, T::MUL_WAVES) void mul_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ ws,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status, uint32_t flags,
                                                    Jac<typename T::F>* __restrict__ park) {
    using F = typename T::F;
    const Header* h = reinterpret_cast<const Header*>(ws);
    const auto* tab = reinterpret_cast<const Entry<F, T::P::NI>*>(ws + HDR_BYTES + chain_bytes<T>());
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const size_t osz = T::out_size(flags);
    uint8_t* o = out + osz * idx;
    const uint32_t st = h->status;
    if (st) {  // the reference's UnmarshalBinary fails once, for every coefficient
        uint32_t* q = reinterpret_cast<uint32_t*>(o);
        for (size_t i = 0; i < osz / 4; i++) q[i] = 0;
        if (status) status[idx] = (uint8_t)st;
        return;
    }
    Aff<F> a;
    if (h->inf) {
        f_zero(a.x);
        f_zero(a.y);
        a.inf = true;
    } else {
        uint32_t k[8];
        T::scalar(k, scalars + 32 * idx);
        Jac<F> r;
        mul<typename T::P>(r, k, tab);
        if (park) {
            park[idx] = r;
            return;
        }
        jac_to_aff(a, r);
    }
    T::encode(o, a, flags);
    if (status) status[idx] = 0;
}
// The parked results of mul_kernel -> wire bytes: a lane takes EB consecutive elements, multiplies their Z's up, inverts
// the product once and walks back (3 multiplications per element instead of an inversion each).  An element at infinity
// (Z = 0) enters the product as one.
template <class F>
constexpr int encode_batch() { return sizeof(F) > 64 ? 4 : 8; }
template <class T>
__global__ __launch_bounds__(64, KYB_TU_WAVES) void encode_kernel(size_t n, const uint8_t* __restrict__ ws, const Jac<typename T::F>* __restrict__ park,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status, uint32_t flags) {
    using F = typename T::F;
    constexpr int EB = encode_batch<F>();
    const Header* h = reinterpret_cast<const Header*>(ws);
    if (h->status || h->inf) return;  // mul_kernel wrote those outputs itself
    const size_t lo = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * EB;
    if (lo >= n) return;
    const int m = (int)(n - lo < (size_t)EB ? n - lo : (size_t)EB);
    F pre[EB];  // pre[j] = Z_0 .. Z_j (infinite elements skipped)
    F run;
    f_one(run);
#pragma unroll
    for (int j = 0; j < EB; j++) {
        if (j < m) {
            const F z = park[lo + j].Z;
            if (!f_is_zero(z)) f_mul(run, run, z);
        }
        pre[j] = run;
    }
    F inv;
    f_inv(inv, run);
    const size_t osz = T::out_size(flags);
#pragma unroll
    for (int j = EB - 1; j >= 0; j--) {
        if (j >= m) continue;
        const Jac<F> r = park[lo + j];
        Aff<F> a;
        a.inf = f_is_zero(r.Z);
        if (a.inf) {
            f_zero(a.x);
            f_zero(a.y);
        } else {
            F zi = inv, zi2;
            if (j > 0) f_mul(zi, inv, pre[j - 1]);  // 1 / Z_j
            f_mul(inv, inv, r.Z);                    // 1 / (Z_0 .. Z_{j-1})
            f_sqr(zi2, zi);
            f_mul(a.x, r.X, zi2);
            f_mul(zi2, zi2, zi);
            f_mul(a.y, r.Y, zi2);
        }
        T::encode(out + osz * (lo + j), a, flags);
        if (status) status[lo + j] = 0;
    }
}

// Enqueue chain (if the base changed) + table + multiplication on `st`.  `key`: the base's wire bytes + flag bytes when
// the caller has them on the host (the host-buffer entry points) -- remembered per (kind, stream) as a HINT that the
// table of that base is (about to be) there, which lets small batches take this path (pairing_abi.cuh); a call without
// it (device pointers) forgets the hint, since it may replace the table.
template <class T>
int run(size_t n, const void* d_scalars, const void* d_base, void* d_out, void* d_status, uint32_t flags, hipStream_t st,
        const std::string* key = nullptr) {
    DeviceCtx* ctx;
    if (int rc = get_ctx(&ctx)) return rc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    void* ws;
    bool grew = false;
    fb_hint_set(ctx, T::KIND, st, nullptr);
    const bool parked = n >= PARK_MIN;
    const size_t tab_bytes = (ws_bytes<T>() + 255) & ~size_t(255);
    if (int rc = ctx_workspace(ctx, T::KIND, st, tab_bytes + (parked ? n * sizeof(Jac<typename T::F>) : 0), &ws, &grew)) return rc;
    if (grew) KYB_HIP_CHECK(hipMemsetAsync(ws, 0, HDR_BYTES, st));
    auto* park = parked ? reinterpret_cast<Jac<typename T::F>*>((uint8_t*)ws + tab_bytes) : nullptr;
    bool rows_chain = false;
#if defined(KYB_ROWFP_INCLUDED)
    if constexpr (HasRowChain<T>::value) {
        static const bool lanes_chain = [] {  // KYB_FB_CHAIN=lanes: the four-lane kernel (A/B)
            const char* e = getenv("KYB_FB_CHAIN");
            return e && e[0] == 'l';
        }();
        if (!lanes_chain) {
            rows_chain = true;
            hipLaunchKernelGGL(chain_rows_kernel<T>, dim3(1), dim3(64), 0, st, (uint8_t*)ws, (const uint8_t*)d_base, flags);
        }
    }
#endif
    if (!rows_chain) hipLaunchKernelGGL(chain_kernel<T>, dim3(1), dim3(4), 0, st, (uint8_t*)ws, (const uint8_t*)d_base, flags);
    hipLaunchKernelGGL(table_kernel<T>, dim3((T::P::NW * NENT + 63) / 64), dim3(64), 0, st, (uint8_t*)ws);
    if (T::needs_member(flags)) hipLaunchKernelGGL(member_kernel<T>, dim3(1), dim3(64), 0, st, (uint8_t*)ws);
    hipLaunchKernelGGL(mul_kernel<T>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, (const uint8_t*)d_scalars, (const uint8_t*)ws,
                       (uint8_t*)d_out, (uint8_t*)d_status, flags, park);
    if (parked) {
        constexpr int EB = encode_batch<typename T::F>();
        const size_t lanes = (n + EB - 1) / EB;
        hipLaunchKernelGGL(encode_kernel<T>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n, (const uint8_t*)ws,
                           (const Jac<typename T::F>*)park, (uint8_t*)d_out, (uint8_t*)d_status, flags);
    }
    if (hipGetLastError() != hipSuccess) {
        hipMemsetAsync(ws, 0, HDR_BYTES, st);  // never trust a half-built table
        set_error("fixed-base multiplication: launch failed");
        return KYB_E_HIP;
    }
    fb_hint_set(ctx, T::KIND, st, key);
    return KYB_OK;
}
#endif  // __HIPCC__

}  // namespace fb
}  // namespace kyb
