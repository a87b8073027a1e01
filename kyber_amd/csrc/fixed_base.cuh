// Fixed-base scalar multiplication for the same-base batches of the pairing suites: out_i = k_i * P for ONE point P and
// many scalars -- share.PriPoly.Commit (share/poly.go:143-149: every coefficient times the same base), key generation
// (x * G2.Base()), `Point.Mul(s, nil)`.  The reference runs its variable-base ladder n times; with the base shared, a
// table of its multiples turns a multiplication into 26 table additions and NO doublings:
//
//   k = sum_w d_w 1024^w, signed digits d_w in [-511, 512] (w = 0..25, the last one holding the top bits + carry), and
//   T[w][j] = (j + 1) 1024^w P in affine form (512 entries per window), so  k P = sum_w sign(d_w) T[w][|d_w| - 1]:
//   26 mixed additions in XYZZ form (curve.cuh, 8M + 2S each) + one inversion, about a sixth of the GLV ladder's field
//   multiplications and a tenth of the plain ladder's (bn256 G2).  (Radix 256 -- 33 additions, a quarter of the table
//   -- was the first version: 6.6 ms per 2^20 BLS12-381 G1 scalars against this one's figure in DESIGN.md.)  No endomorphism is involved, so the result is the
//   integer multiple for EVERY decodable base (bn256's G2 points outside the order-n subgroup included) and every
//   256-bit scalar, like the reference's double-and-add.
//
// The table costs a chain of 250 dependent doublings (one lane: ~2.5 ms on G1, three times that on G2) + 26 x 512
// independent small multiples: it pays for itself from ~2^17 scalars, or at any batch size once it exists -- the
// workspace keeps the last base's table per (suite, group, stream), and the chain kernel recognises the base on the
// device (no host round trip).  Table: 26 x 512 x (2 field elements) = 1.3 MB (BLS12-381 G1) .. 2.6 MB (G2): inside the
// 4 MB of L2 each XCD has.
#pragma once
#include "coop_slots.cuh"
#include "curve.cuh"
#if defined(__HIPCC__)
#include "context.h"
#endif

namespace kyb {
namespace fb {

constexpr int WBITS = 10, NWIN = (256 + WBITS) / WBITS, NENT = 1 << (WBITS - 1), WIRE_MAX = 192;
static_assert(NWIN * WBITS >= 257, "the last window must hold the recoding's carry");
constexpr uint64_t MAGIC = 0x6b79626662763032ull;  // "kybfbv02"

// head of the workspace: which base the table below belongs to
struct Header {
    uint64_t magic;
    uint32_t key_flags;   // the decode-relevant flag bits the base was decoded under
    uint32_t key_len;     // bytes of its wire form
    uint32_t status;      // UnmarshalBinary's verdict on it (ST_OK = 0)
    uint32_t inf;         // it is the point at infinity
    uint32_t fresh;       // set by the chain kernel of THIS call: the table has to be rebuilt
    uint32_t pad;
    uint8_t key[WIRE_MAX];
};

template <class F>
struct Entry {  // affine multiple; (0, 0) -- not a point of y^2 = x^3 + b, b != 0 -- stands for infinity
    F x, y;
};

// window w of k: bits [WBITS w, WBITS w + WBITS) of the 256-bit integer (zero beyond bit 255)
KYB_HD int window_bits(const uint32_t (&k)[8], int w) {
    const int bit = w * WBITS, idx = bit >> 5, sh = bit & 31;
    if (idx >= 8) return 0;
    uint32_t v = k[idx] >> sh;
    if (sh + WBITS > 32 && idx + 1 < 8) v |= k[idx + 1] << (32 - sh);
    return (int)(v & ((1u << WBITS) - 1));
}
// signed radix-2^WBITS digits: d[w] in [-(NENT - 1), NENT]
KYB_HD void digits(int (&d)[NWIN], const uint32_t (&k)[8]) {
    int carry = 0;
#pragma unroll
    for (int w = 0; w < NWIN; w++) {
        const int v = window_bits(k, w) + carry;
        carry = v > NENT ? 1 : 0;
        d[w] = v - (carry << WBITS);
    }
}

// q[w] = 2^(WBITS w) P  (Jacobian): the one serial part of the table
template <class F>
KYB_HD void chain(Jac<F> (&q)[NWIN], const Aff<F>& base) {
    jac_from_aff(q[0], base);
#pragma unroll 1
    for (int w = 1; w < NWIN; w++) {
        Jac<F> t = q[w - 1];
#pragma unroll 1
        for (int i = 0; i < WBITS; i++) jac_dbl_inl(t, t);
        q[w] = t;
    }
}

// e = (j + 1) Q in affine form, j in [0, NENT)
template <class F>
KYB_HD void entry(Entry<F>& e, const Jac<F>& q, int j) {
    const int m = j + 1;  // 1 .. NENT
    Jac<F> acc;
    jac_set_inf(acc);
#pragma unroll 1
    for (int bit = WBITS - 1; bit >= 0; bit--) {
        jac_dbl_inl(acc, acc);
        if ((m >> bit) & 1) jac_add_inl<F, false>(acc, acc, q);
    }
    Aff<F> a;
    jac_to_aff(a, acc);
    e.x = a.x;
    e.y = a.y;
    if (a.inf) {
        f_zero(e.x);
        f_zero(e.y);
    }
}

// r = k * P from P's table (tab[w * NENT + j]); r comes back in Jacobian form
template <class F>
KYB_HD void mul(Jac<F>& r, const uint32_t (&k)[8], const Entry<F>* __restrict__ tab) {
    Xyzz<F> acc;
    xyzz_set_inf(acc);
    int carry = 0;
#pragma unroll 1
    for (int w = 0; w < NWIN; w++) {
        // the digits of digits(), produced as the walk needs them
        const int v = window_bits(k, w) + carry;
        carry = v > NENT ? 1 : 0;
        const int dw = v - (carry << WBITS);
        if (dw == 0) continue;
        const int a = dw < 0 ? -dw : dw;
        const Entry<F> e = tab[w * NENT + (a - 1)];
        if (f_is_zero(e.x) & f_is_zero(e.y)) continue;  // that multiple of the base is the point at infinity
        F y = e.y, ny;
        f_neg(ny, e.y);
        f_cmov(y, ny, dw < 0);
        xyzz_madd(acc, e.x, y);
    }
    xyzz_to_jac(r, acc);
}


#if defined(__HIPCC__)
// ---- kernels and the enqueue routine, over a per-(suite, group) traits type T:
//   F; decode(Aff<F>&, wire, flags) -> status; encode(out, Aff<F>, flags); wire_size(flags); out_size(flags);
//   scalar(k[8], wire32); KIND (workspace kind); KEY_FLAGS (the flag bits decode() looks at)
constexpr size_t HDR_BYTES = 512;
template <class T>
constexpr size_t chain_bytes() { return ((sizeof(Jac<typename T::F>) * NWIN + 255) / 256) * 256; }
template <class T>
constexpr size_t ws_bytes() { return HDR_BYTES + chain_bytes<T>() + sizeof(Entry<typename T::F>) * NWIN * NENT; }

// One lane: is the table already this base's?  Otherwise decode the base and walk the doubling chain.
template <class T>
__global__ __launch_bounds__(64) void chain_kernel(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using F = typename T::F;
    Header* h = reinterpret_cast<Header*>(ws);
    const uint32_t kf = flags & T::KEY_FLAGS, len = (uint32_t)T::wire_size(flags);
    bool same = h->magic == MAGIC && h->key_flags == kf && h->key_len == len;
    for (uint32_t i = 0; i < len && same; i++) same = h->key[i] == base[i];
    if (same) {
        h->fresh = 0;
        return;
    }
    h->magic = 0;
    Aff<F> a;
    const int st = T::decode(a, base, flags);
    h->status = (uint32_t)st;
    h->inf = (st == 0 && a.inf) ? 1u : 0u;
    h->key_flags = kf;
    h->key_len = len;
    for (uint32_t i = 0; i < len; i++) h->key[i] = base[i];
    h->fresh = 1;
    if (st == 0 && !a.inf) {
        Jac<F>* q = reinterpret_cast<Jac<F>*>(ws + HDR_BYTES);
        Jac<F> t;
        jac_from_aff(t, a);
        q[0] = t;
#pragma unroll 1
        for (int w = 1; w < NWIN; w++) {
#pragma unroll 1
            for (int i = 0; i < WBITS; i++) jac_dbl_inl(t, t);
            q[w] = t;
        }
    }
    __threadfence();
    h->magic = MAGIC;
}
// The same on FOUR cooperating lanes (coop_slots.cuh: a doubling is three product levels deep instead of seven
// multiplications) -- NOT the default: written at the end of round 3 without GPU time left to measure it, selected by
// KYB_FB_CHAIN=coop for the A/B run that decides (the slot arithmetic itself runs in the MSM's reduce kernel and, on the
// CPU, with threads as lanes).  One workgroup of four lanes; lane 0 does what chain_kernel's lane does before and after
// the doublings.
template <class T>
__global__ __launch_bounds__(64, 2) void chain_coop_kernel(uint8_t* __restrict__ ws, const uint8_t* __restrict__ base, uint32_t flags) {
    using F = typename T::F;
    constexpr int P = 0, TMP = 3;
    __shared__ coop::Slot<F> S[TMP + coop::TEMPS];
    __shared__ int go;
    const int r = (int)threadIdx.x & 3;
    Header* h = reinterpret_cast<Header*>(ws);
    Jac<F>* q = reinterpret_cast<Jac<F>*>(ws + HDR_BYTES);
    if (threadIdx.x == 0) {
        const uint32_t kf = flags & T::KEY_FLAGS, len = (uint32_t)T::wire_size(flags);
        bool same = h->magic == MAGIC && h->key_flags == kf && h->key_len == len;
        for (uint32_t i = 0; i < len && same; i++) same = h->key[i] == base[i];
        go = 0;
        if (same) {
            h->fresh = 0;
        } else {
            h->magic = 0;
            Aff<F> a;
            const int st = T::decode(a, base, flags);
            h->status = (uint32_t)st;
            h->inf = (st == 0 && a.inf) ? 1u : 0u;
            h->key_flags = kf;
            h->key_len = len;
            for (uint32_t i = 0; i < len; i++) h->key[i] = base[i];
            h->fresh = 1;
            if (st == 0 && !a.inf) {
                Jac<F> t;
                jac_from_aff(t, a);
                q[0] = t;
                S[P].f = t.X;
                S[P + 1].f = t.Y;
                S[P + 2].f = t.Z;
                go = 1;
            } else {
                __threadfence();
                h->magic = MAGIC;
            }
        }
    }
    __syncthreads();
    if (!go) return;  // uniform: the table is current, or there is nothing to build
#pragma unroll 1
    for (int w = 1; w < NWIN; w++) {
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
// One lane per table entry
template <class T>
__global__ __launch_bounds__(64) void table_kernel(uint8_t* __restrict__ ws) {
    using F = typename T::F;
    const Header* h = reinterpret_cast<const Header*>(ws);
    if (!h->fresh || h->status || h->inf) return;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= NWIN * NENT) return;
    const int w = t / NENT, j = t - w * NENT;
    const Jac<F> q = reinterpret_cast<const Jac<F>*>(ws + HDR_BYTES)[w];
    Entry<F> e;
    entry(e, q, j);
    reinterpret_cast<Entry<F>*>(ws + HDR_BYTES + chain_bytes<T>())[t] = e;
}
// One lane per scalar
template <class T>
__global__ __launch_bounds__(64, 2) void mul_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ ws,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status, uint32_t flags) {
    using F = typename T::F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const Header* h = reinterpret_cast<const Header*>(ws);
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
        mul(r, k, reinterpret_cast<const Entry<F>*>(ws + HDR_BYTES + chain_bytes<T>()));
        jac_to_aff(a, r);
    }
    T::encode(o, a, flags);
    if (status) status[idx] = 0;
}

// Enqueue chain (if the base changed) + table + multiplication on `st`
template <class T>
int run(size_t n, const void* d_scalars, const void* d_base, void* d_out, void* d_status, uint32_t flags, hipStream_t st) {
    DeviceCtx* ctx;
    if (int rc = get_ctx(&ctx)) return rc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    void* ws;
    bool grew = false;
    if (int rc = ctx_workspace(ctx, T::KIND, st, ws_bytes<T>(), &ws, &grew)) return rc;
    if (grew) KYB_HIP_CHECK(hipMemsetAsync(ws, 0, HDR_BYTES, st));
    static const bool coop_chain = [] {
        const char* e = getenv("KYB_FB_CHAIN");
        return e && e[0] == 'c';
    }();
    if (coop_chain) hipLaunchKernelGGL(chain_coop_kernel<T>, dim3(1), dim3(4), 0, st, (uint8_t*)ws, (const uint8_t*)d_base, flags);
    else hipLaunchKernelGGL(chain_kernel<T>, dim3(1), dim3(64), 0, st, (uint8_t*)ws, (const uint8_t*)d_base, flags);
    hipLaunchKernelGGL(table_kernel<T>, dim3((NWIN * NENT + 63) / 64), dim3(64), 0, st, (uint8_t*)ws);
    hipLaunchKernelGGL(mul_kernel<T>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, (const uint8_t*)d_scalars,
                       (const uint8_t*)ws, (uint8_t*)d_out, (uint8_t*)d_status, flags);
    if (hipGetLastError() != hipSuccess) {
        hipMemsetAsync(ws, 0, HDR_BYTES, st);  // never trust a half-built table
        set_error("fixed-base multiplication: launch failed");
        return KYB_E_HIP;
    }
    return KYB_OK;
}
#endif  // __HIPCC__

}  // namespace fb
}  // namespace kyb
