// Entry into the cooperative tower machine for BLS12-381 (bls12381_pair.hip) from the other translation units:
// the fused BLS verification (bls12381_h2c.hip) prepares its operands per lane and runs the same CHECK program.
#pragma once
#include "context.h"

namespace kyb {
namespace blsvm {
constexpr int FP_WORDS = 12;          // packed Montgomery words of the per-lane field code (mont.cuh)
constexpr int PAIR_INPUTS = 6;        // P.x, P.y, Q.x.c0, Q.x.c1, Q.y.c0, Q.y.c1
constexpr int CHECK_INPUTS = 12;      // pair A, then pair B with P_B already negated
// flag byte per pairing: bit 0 pair A has an operand at infinity, bit 1 pair B, bit 7 an input was rejected
constexpr uint8_t FL_DEAD_A = 1, FL_DEAD_B = 2, FL_REJECTED = 0x80;

// Workspace of one call: operand arrays [input][n][FP_WORDS], the flag bytes, and the machine's global scratch.
struct Work {
    uint32_t* in;
    uint8_t* flags;
    uint32_t* gspill;
    unsigned grid;
};
// Takes the (WS_PAIR, stream) workspace for n pairings with `ninputs` operands each.  The caller holds ctx->enq_mu
// from here until the machine is enqueued.
int workspace(DeviceCtx* ctx, hipStream_t st, size_t n, int ninputs, Work* w);
// Enqueue the PAIR program: GT bytes (576 per pairing) from the operands in `w`.
int launch_pair(const Work& w, size_t n, uint8_t* d_gt, hipStream_t st);
// Enqueue the CHECK program: one boolean per pairing.
int launch_check(const Work& w, size_t n, uint8_t* d_ok, hipStream_t st);
}  // namespace blsvm
}  // namespace kyb
