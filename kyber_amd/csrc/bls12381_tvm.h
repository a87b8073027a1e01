// Entry into the cooperative tower machine for BLS12-381 (bls12381_pair.hip) from the other translation units:
// the fused BLS verification (bls12381_h2c.hip) prepares its operands per lane and runs the same CHECK program.
#pragma once
#include "context.h"

namespace kyb {
namespace blsvm {
constexpr int FP_WORDS = 12;          // packed Montgomery words of the per-lane field code (mont.cuh)
constexpr int PAIR_INPUTS = 6;        // P.x, P.y, Q.x.c0, Q.x.c1, Q.y.c0, Q.y.c1
constexpr int CHECK_INPUTS = 12;      // pair A, then pair B with P_B already negated

// One point argument of a batch call, prepared by its own lanes (bls12381_prep.hip): n lanes per operand, so that the
// G1 and G2 decompressions, subgroup checks and hashes of one pairing run side by side instead of one after the other
// in a single lane (65 536 pairings are only one wave per SIMD; four operands are four).
// OPND_STATUS: an operand every lane shares (the one public key of a same-key verification): no coordinates, its status
// byte -- src[0], on the device -- is copied to every pairing
// OPND_G1_SHARED_HASH: ONE message for the whole batch (sign/tbls/tbls.go:118-131: every partial signature is over the same
// msg): src = the message, stride = its length.  One extra workgroup of the operand kernel hashes it once into
// Work::shared while the other operands decode; launch_prep's second kernel deals the point to every pairing.
enum OperandKind : uint32_t { OPND_G1 = 0, OPND_G2 = 1, OPND_G1_HASH = 2, OPND_G2_HASH = 3, OPND_G1_GEN = 4, OPND_G2_GEN = 5, OPND_STATUS = 6,
                              OPND_G1_SHARED_HASH = 7 };
struct Operand {
    const uint8_t* src;  // wire encodings (messages for the hash kinds; unused for the generators)
    uint32_t kind;
    uint32_t stride;     // bytes per element of src (hash kinds: the message length)
    uint32_t first;      // first input index it fills: a G1 point takes 2 field elements, a G2 point 4
    uint32_t negate;     // store -P (the second pair of a product check)
    uint32_t arg;        // position among the call's point arguments (KYB_F_TRUSTED bit) for the decode kinds
};
constexpr int MAX_OPERANDS = 4;
// status byte per (operand, pairing): bits 0-6 the UnmarshalBinary status, bit 7 the point is at infinity.  The
// machine derives its lane flags from them: operands 0, 1 form pair A, operands 2, 3 pair B.
constexpr uint8_t PST_INF = 0x80;

// Workspace of one call: operand arrays [input][n][FP_WORDS], the per-operand status bytes, the machine's global scratch.
struct Work {
    uint32_t* in;
    uint8_t* pst;
    uint32_t* gspill;
    uint32_t* shared;  // 256 bytes: the affine coordinates (2 x FP_WORDS) of an operand every pairing shares
    unsigned grid;
    uint32_t g2_member;  // set by launch_prep: bit k = operand k is a G2 point decoded WITHOUT its r-torsion test, which
                         // the machine's program decides at the end of the Miller loop (tvm::Args::g2_member)
};
// Enqueue the operand kernel: ops[0 .. nops) -> w.in / w.pst.  dst: the hash-to-curve tag of the hash kinds.
int launch_prep(Work& w, size_t n, const Operand* ops, int nops, uint32_t flags, const uint8_t* dst, size_t dst_len,
                hipStream_t st);
// Takes the (WS_PAIR, stream) workspace for n pairings with `ninputs` operands each.  The caller holds ctx->enq_mu
// from here until the machine is enqueued.
int workspace(DeviceCtx* ctx, hipStream_t st, size_t n, int ninputs, Work* w);
// Enqueue the PAIR program: GT bytes (576 per pairing) from the operands in `w`; d_status (may be null) receives the
// first non-zero operand status of every pairing.
int launch_pair(const Work& w, size_t n, uint8_t* d_gt, uint8_t* d_status, hipStream_t st);
// Enqueue the CHECK program: one boolean per pairing.
int launch_check(const Work& w, size_t n, uint8_t* d_ok, uint8_t* d_status, hipStream_t st);
// Enqueue the VERIFY program: CHECK whose second G2 operand is the generator (its Miller lines are a table of
// constants); operands 0, 1 = pair A (G1, G2), operand 2 = pair B's G1 point (negated), inputs 0-7.
constexpr int VERIFY_INPUTS = 8;
int launch_verify(const Work& w, size_t n, uint8_t* d_ok, uint8_t* d_status, hipStream_t st);
// Same-key verification (sign/bls/bls.go:82-96 for many messages under ONE public key): program VERIFYK takes the lines
// of BOTH Miller loops from a table [generator | key].  prepare_key enqueues the kernel that decodes the key (d_key: its
// wire form on the device) with UnmarshalBinary's rules and, unless the (WS_VKEY, stream) workspace already holds this
// key's table, walks it through the loop once (bls12381_keylines.cuh); *table / *kst: the table and the key's status
// byte (PST_INF convention), valid for kernels enqueued after it on `st`.
constexpr int VERIFYK_INPUTS = 4;  // H(m) (2), -sig (2)
int prepare_key(DeviceCtx* ctx, hipStream_t st, const uint8_t* d_key, uint32_t flags, const int32_t** table, const uint8_t** kst);
// operands 0 (H(m)), 1 (the key: OPND_STATUS), 2 (the signature, negated; inputs 2-3)
int launch_verify_same_key(const Work& w, size_t n, const int32_t* table, uint8_t* d_ok, uint8_t* d_status, hipStream_t st);
}  // namespace blsvm
}  // namespace kyb
