// Operand kernel of the BLS12-381 pairing calls: ONE POINT ARGUMENT PER LANE.  Every operand of a batch call --
// a G1 or G2 encoding to unmarshal with the reference's checks (kilic/g1.go:127-131: ZCash flags, on-curve, subgroup),
// a message to hash to the curve (kilic/g1.go:161-170, RFC 9380), or the group generator -- gets n lanes of its own
// and leaves affine coordinates (packed Montgomery words) + one status byte in the tower machine's workspace.
//
// Round 2's first version decoded all operands of a pairing in one lane: 65 536 pairings are 1024 waves, one per SIMD,
// each running a G1 square root + subgroup check and then a G2 one (5.1 ms, 18 % of a checked Pair call; 10.2 ms for a
// check).  With a lane per operand the same batch is 2048 - 4096 waves, a small batch reaches the whole chip sooner,
// and the operands of one pairing no longer queue behind each other.  Measured at 65 536 pairings the kernel is
// VALU-bound either way (about 2.0e6 field-arithmetic instructions per pairing: two square roots, 126 + 63 doublings of
// the subgroup checks): 5.0 / 10.1 ms with the default one-wave register budget, 4.7 / 10.8 ms with a two-wave one.
#include "bls12381_h2c.cuh"
#include "bls12381_tvm.h"
#include "pairing_abi.cuh"

#include <string.h>

namespace kyb {
namespace blsvm {

struct PrepArgs {
    Operand op[MAX_OPERANDS];
    int nops;
    size_t n;
    uint32_t* in;
    uint8_t* pst;
    uint32_t* shared;
    int has_shared;  // a workgroup ahead of the operands' own hashes the OPND_G1_SHARED_HASH message
    uint32_t flags;
    bls::DstArg dst;
};

__device__ __forceinline__ void put_fp(uint32_t* in, size_t n, uint32_t idx, size_t i, const bls::fp& x) {
    uint32_t* d = in + ((size_t)idx * n + i) * FP_WORDS;
#pragma unroll
    for (int k = 0; k < FP_WORDS; k += 4) *reinterpret_cast<uint4*>(d + k) = make_uint4(x.v[k], x.v[k + 1], x.v[k + 2], x.v[k + 3]);
}

__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_operand_kernel(PrepArgs a) {
    const size_t nblk = (a.n + 63) / 64;
    size_t blk = blockIdx.x;
    if (a.has_shared) {
        // workgroup 0 (dispatched first: its one lane is the longest chain of the launch) hashes the batch's shared
        // message once, next to the lanes that decode the other operands, at raised issue priority
        if (blk == 0) {
            if (threadIdx.x != 0) return;
            __builtin_amdgcn_s_setprio(3);
            for (int j = 0; j < a.nops; j++) {
                const Operand& o = a.op[j];
                if (o.kind != OPND_G1_SHARED_HASH) continue;
                bls::g1_jac h;
                bls::g1_aff p;
                bls::hash_g1_point(h, o.src, o.stride, a.dst);
                jac_to_aff(p, h);
                if (o.negate) fp_neg(p.y, p.y);
#pragma unroll
                for (int w = 0; w < FP_WORDS; w++) {
                    a.shared[w] = p.x.v[w];
                    a.shared[FP_WORDS + w] = p.y.v[w];
                }
                a.shared[2 * FP_WORDS] = p.inf ? 1u : 0u;
            }
            return;
        }
        blk -= 1;
    }
    const int k = (int)(blk / nblk);  // uniform per workgroup: no divergence between the kinds
    const size_t i = (blk - (size_t)k * nblk) * 64 + threadIdx.x;
    if (i >= a.n) return;
    const Operand& o = a.op[k];
    if (o.kind == OPND_G1_SHARED_HASH) return;  // dealt out by bls12381_shared_operand_kernel once the hash is there
    if (o.kind == OPND_STATUS) {  // an operand shared by every pairing: only its verdict
        a.pst[(size_t)k * a.n + i] = o.src[0];
        return;
    }
    int st = bls::ST_OK;
    bool inf = false;
    if (o.kind == OPND_G1 || o.kind == OPND_G1_HASH || o.kind == OPND_G1_GEN) {
        bls::g1_aff p;
        if (o.kind == OPND_G1) {
            st = bls::g1_decode_f(p, o.src + (size_t)o.stride * i, a.flags, (int)o.arg);
        } else if (o.kind == OPND_G1_HASH) {
            bls::g1_jac h;
            bls::hash_g1_point(h, o.src + (size_t)o.stride * i, o.stride, a.dst);
            jac_to_aff(p, h);
        } else {
            bls::fp_const(p.x, bls::CC::G1X);
            bls::fp_const(p.y, bls::CC::G1Y);
            p.inf = false;
        }
        if (o.negate) fp_neg(p.y, p.y);
        inf = p.inf;
        put_fp(a.in, a.n, o.first, i, p.x);
        put_fp(a.in, a.n, o.first + 1, i, p.y);
    } else {
        bls::g2_aff q;
        if (o.kind == OPND_G2) {
            // an unvouched-for operand: every rule of UnmarshalBinary here except the r-torsion test, which the machine
            // reads off the end of its Miller loop (gen_tower_vm.py bls_g2_member_check) -- 63 doublings saved per point
            st = flag_trusted(a.flags, (int)o.arg) ? bls::g2_decode_f(q, o.src + (size_t)o.stride * i, a.flags, (int)o.arg)
                                                        : bls::g2_decode_on_curve(q, o.src + (size_t)o.stride * i, a.flags);
        } else if (o.kind == OPND_G2_HASH) {
            bls::g2_jac h;
            bls::hash_g2_point(h, o.src + (size_t)o.stride * i, o.stride, a.dst);
            jac_to_aff(q, h);
        } else {
            fp2_load_const<bls::TC>(q.x, bls::CC::G2X);
            fp2_load_const<bls::TC>(q.y, bls::CC::G2Y);
            q.inf = false;
        }
        if (o.negate) fp2_neg(q.y, q.y);
        inf = q.inf;
        put_fp(a.in, a.n, o.first, i, q.x.c0);
        put_fp(a.in, a.n, o.first + 1, i, q.x.c1);
        put_fp(a.in, a.n, o.first + 2, i, q.y.c0);
        put_fp(a.in, a.n, o.first + 3, i, q.y.c1);
    }
    a.pst[(size_t)k * a.n + i] = (uint8_t)((st & 0x7f) | ((inf && st == bls::ST_OK) ? PST_INF : 0));
}

// the shared operand's coordinates and status to every pairing (after the operand kernel: stream order)
__global__ __launch_bounds__(256) void bls12381_shared_operand_kernel(size_t n, const uint32_t* __restrict__ shared, uint32_t* __restrict__ in,
                                                                      uint8_t* __restrict__ pst, uint32_t first, int k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        uint32_t* d = in + ((size_t)(first + c) * n + i) * FP_WORDS;
#pragma unroll
        for (int w = 0; w < FP_WORDS; w += 4)
            *reinterpret_cast<uint4*>(d + w) = *reinterpret_cast<const uint4*>(shared + c * FP_WORDS + w);
    }
    pst[(size_t)k * n + i] = shared[2 * FP_WORDS] ? PST_INF : 0;
}

int launch_prep(Work& w, size_t n, const Operand* ops, int nops, uint32_t flags, const uint8_t* dst, size_t dst_len,
                hipStream_t st) {
    if (nops < 1 || nops > MAX_OPERANDS || dst_len > 255 || (dst_len && !dst)) {
        set_error("pairing operands: bad argument");
        return KYB_E_ARG;
    }
    PrepArgs a;
    memset(&a, 0, sizeof a);
    w.g2_member = 0;
    for (int k = 0; k < nops; k++) {
        a.op[k] = ops[k];
        if (ops[k].kind == OPND_G2 && !flag_trusted(flags, (int)ops[k].arg)) w.g2_member |= 1u << k;
    }
    a.nops = nops;
    a.n = n;
    a.in = w.in;
    a.pst = w.pst;
    a.shared = w.shared;
    a.flags = flags;
    if (dst_len) memcpy(a.dst.b, dst, dst_len);
    a.dst.len = (uint32_t)dst_len;
    const size_t nblk = (n + 63) / 64;
    int shared_k = -1;
    for (int k = 0; k < nops; k++)
        if (ops[k].kind == OPND_G1_SHARED_HASH) shared_k = k;
    a.has_shared = shared_k >= 0 ? 1 : 0;
    hipLaunchKernelGGL(bls12381_operand_kernel, dim3((unsigned)(nblk * nops + (shared_k >= 0 ? 1 : 0))), dim3(64), 0, st, a);
    if (shared_k >= 0)
        hipLaunchKernelGGL(bls12381_shared_operand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const uint32_t*)w.shared,
                           w.in, w.pst, ops[shared_k].first, shared_k);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}

}  // namespace blsvm
}  // namespace kyb
