/*
 * kyber_hip.h -- C ABI of libkyberhip.so, the MI355X (gfx950) batched
 * group-arithmetic engine that sits behind dedis/kyber's kyber.Point.Mul /
 * pairing.Suite hot path.
 *
 * This is what a cgo binding in the reference would import (see INTEGRATION.md
 * for the Go stub).  Plain pointers and sizes only.
 *
 * Conventions
 *   - every entry point returns 0 on success, <0 on a call-level error
 *     (KYB_E_*); kyb_last_error() gives a message for the calling thread.
 *   - wire formats are exactly the reference's MarshalBinary encodings;
 *     element-major, densely packed.
 *   - where decoding an input can fail (UnmarshalBinary returning an error in
 *     the reference) the call takes `status[n]`: 0 = ok, KYB_ST_* otherwise,
 *     and the corresponding output element is all-zero bytes.
 *   - "*_dev" variants take DEVICE pointers (already resident in HBM) and a
 *     hipStream_t passed as void* (NULL = default stream); they only enqueue
 *     work and never synchronise.  The host variants copy in, run, copy out
 *     and synchronise.
 *   - all entry points are thread-safe; state is a per-device context that is
 *     created on first use on the calling thread's current HIP device.
 *
 * TIMING: BY DEFAULT NOTHING IN THIS LIBRARY IS CONSTANT-TIME.  Results equal the
 * reference's on every input, but window tables are indexed by scalar digits
 * (through L2), exceptional additions branch, field inversion is a
 * data-dependent loop and KYB_F_VARTIME skips leading zero windows.  The one
 * exception is opt-in: KYB_F_UNIFORM on the Ed25519 multiplications (below).  A suite
 * built on it belongs in suites/all_vartime.go only and must not be offered
 * where suites.RequireConstantTime (suites/suites.go:67) is expected to hold
 * (the reference's default Ed25519 Mul scans its table with CMove,
 * group/edwards25519/ge.go:419-435).  See INTEGRATION.md section 2b.
 */
#ifndef KYBER_HIP_H
#define KYBER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KYB_OK 0
#define KYB_E_ARG (-1)    /* bad argument (NULL pointer, bad flags) */
#define KYB_E_HIP (-2)    /* a HIP runtime call failed */
#define KYB_E_NODEV (-3)  /* no usable gfx950 device */
#define KYB_E_ALLOC (-4)  /* workspace allocation failed */

#define KYB_ST_OK 0
#define KYB_ST_BAD_POINT 1 /* encoding is not a curve point (reference: UnmarshalBinary error) */
#define KYB_ST_NOT_IN_SUBGROUP 2

/* flags */
#define KYB_F_VARTIME 1u /* Ed25519: geScalarMultVartime semantics (all 256 scalar bits honoured,
                            group/edwards25519/ge_mult_vartime.go:11) and leading zero
                            windows skipped per wave. Default = the VALUE semantics of the reference's
                            constant-time path ge.go:443 incl. its >= 2^255 behaviour (not its timing). */

#define KYB_F_UNIFORM 8u /* Ed25519 mul_base / mul / mul_same_base: scalar-INDEPENDENT memory addresses and control flow --
                            the access pattern of the reference's default (constant-time) Mul, group/edwards25519/ge.go:352-371,
                            419-435: every window reads ALL eight table candidates and keeps one by mask, the sign is a
                            conditional move, no window is skipped; same value semantics as the default (incl. >= 2^255).
                            For secret scalars: PriPoly.Commit's coefficients (share/poly.go:143-149), signing nonces
                            (sign/eddsa/eddsa.go), DH.  mul_same_base builds the base's table at any batch size.  Costs
                            ~2x on the fixed-base path (64 additions instead of 32) -- measured, DESIGN.md.  Exclusive with
                            KYB_F_VARTIME.  It removes the digit-indexed loads and the scalar-dependent schedule; the
                            library makes no formal constant-time claim (see TIMING above). */

/* Pairing-suite calls (trailing `flags` argument of mul / msm / pair / pair_check / verify):
 *  KYB_F_UNCOMPRESSED  BLS12-381 point INPUTS are ZCash uncompressed affine (G1 96 B x||y, G2 192 B
 *                      x.c1||x.c0||y.c1||y.c0): no square root (the `_aff` form of SURVEY.md 8b).  Outputs stay
 *                      compressed.  bn256's wire format is already uncompressed: the flag changes nothing there.
 *  KYB_F_TRUSTED(i)    point argument i (0-based, in declaration order) holds points that were validated before --
 *                      outputs of this library or of an earlier UnmarshalBinary, as every kyber.Point handed to
 *                      Mul / Pair already is in the reference (validation happens once, in UnmarshalBinary,
 *                      kilic/g1.go:127-131) -- so the subgroup check (and, for uncompressed input, the curve-equation
 *                      check) is skipped.  Precondition, not a check: a point outside the subgroup then gives a
 *                      result the reference could never produce.  bn256 has no subgroup check to skip. */
#define KYB_F_UNCOMPRESSED 2u
#define KYB_F_UNCOMPRESSED_OUT 4u /* g1/g2 mul and mul_same_base: write uncompressed outputs (96 / 192 B) as well, so a
                                     pipeline can keep points in the form that needs no square root; bn256: no-op */
#define KYB_F_SCALAR_BITS(b) ((uint32_t)(b) << 16) /* *_msm only: every scalar is below 2^b (1 <= b <= 256) -- the
                                     128-bit coefficients of sign/bdn (bdn.go:29-63) run half the windows.  Bits at
                                     and above b are IGNORED (the result is sum (k_i mod 2^b) P_i).  BLS12-381 G1,
                                     which splits full-length scalars into 127-bit halves, honours it for b <= 160
                                     (plain windows) and takes no notice above. */
#define KYB_F_SCALAR_BITS_MASK (0x1ffu << 16)
#define KYB_F_TRUSTED(i) (0x100u << (i))
#define KYB_F_TRUSTED_ALL 0xF00u /* the four point arguments of pair_check; calls with fewer point arguments reject the extra bits */

int kyb_version(void);
const char *kyb_last_error(void);

/* Number of visible devices; creates nothing. */
int kyb_device_count(void);
/* Eagerly create the context (tables, workspace) for the current HIP device. */
int kyb_init(void);
/* Release every per-device context. */
int kyb_shutdown(void);

/* ---- several GPUs of one node behind the same calls (the reference's callers are single-process
 * loops -- share/poly.go:143-149, sign/bdn/bdn.go:126-161 -- so they need ONE call, not one process per GPU).
 * After kyb_init_devices(ndev) (devices 0 .. ndev-1) or kyb_set_devices(list) every HOST-BUFFER batch entry point of
 * at least kyb_set_shard_threshold() units (default 16384) is cut into `ndev` contiguous slices by the rule of
 * kyb_shard_range (sizes differ by at most one), one host thread + device context per slice, no traffic between
 * devices; kyb_*_msm shards the points, runs the whole Pippenger pipeline per device and adds the ndev encoded partial
 * points at the end (<= 129 bytes per device cross the host, bucket arrays never move).  A device may be listed more
 * than once (its slices then run one after the other; used by the tests on a one-GPU box).  ndev = 0 / an empty list
 * restores single-device behaviour.  The `_dev` entry points are unaffected.  Results do not depend on the device
 * set. */
int kyb_init_devices(int ndev);
int kyb_set_devices(const int *devices, int ndev);
int kyb_get_devices(int *out, int cap); /* returns the number of shards configured (0: single device) */
int kyb_set_shard_threshold(size_t min_units);
void kyb_shard_range(size_t n, int rank, int world, size_t *lo, size_t *hi);
/* The `_dev` entry points keep one grow-only device workspace per (kind of call, stream) so that calls on different
 * streams never share scratch.  A caller that destroys a stream calls this first (current device): waits for the
 * stream and frees the workspaces tied to that handle.  Never needed for long-lived streams or the host-buffer calls. */
int kyb_stream_release(void *stream);

/* ------------------------------------------------------------------ Ed25519
 * scalars: 32-byte little-endian, taken as plain 256-bit integers (never
 * reduced mod l: group/edwards25519/scalar.go:226-233).
 * points : 32-byte compressed (group/edwards25519/ge.go:99-150).           */

/* out[i] = scalars[i] * B.   Replaces (*point).Mul(s, nil) ->
 * geScalarMultBase (group/edwards25519/point.go:243, ge.go:373) + MarshalBinary. */
int kyb_ed25519_mul_base(size_t n, const uint8_t *scalars, uint8_t *out, uint32_t flags);
int kyb_ed25519_mul_base_dev(size_t n, const void *d_scalars, void *d_out, uint32_t flags, void *stream);

/* out[i] = scalars[i] * points[i].   Replaces UnmarshalBinary + (*point).Mul(s, A)
 * -> geScalarMult / geScalarMultVartime (point.go:235-258, ge.go:443,
 * ge_mult_vartime.go:11) + MarshalBinary. */
int kyb_ed25519_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out,
                    uint8_t *status, uint32_t flags);
int kyb_ed25519_mul_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out,
                        void *d_status, uint32_t flags, void *stream);

/* out[i] = scalars[i] * point  (one shared base).  Replaces the loop of
 * share.PriPoly.Commit (share/poly.go:143-149).  If the point does not decode
 * every status[i] is KYB_ST_BAD_POINT and every output is zero. */
int kyb_ed25519_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[32],
                              uint8_t *out, uint8_t *status, uint32_t flags);

/* out[i] = a[i] + b[i]: (*point).Add (group/edwards25519/point.go:216-223 -> ge.go:183), both operands
 * unmarshalled with the reference's rules. */
int kyb_ed25519_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
int kyb_ed25519_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);

/* out[i] = MarshalBinary(UnmarshalBinary(points[i])), status[i] = the error UnmarshalBinary would return:
 * (*point).UnmarshalBinary (group/edwards25519/point.go:65-70 -> ge.go:110-150; bit 255 of y is only the sign of x,
 * y >= p is accepted, no subgroup check) followed by MarshalBinary (point.go:54-58), i.e. the canonical encoding.
 * Rejected slots give 32 zero bytes. */
int kyb_ed25519_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status);
int kyb_ed25519_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, void *stream);

/* out[i] = Hash(msgs[i], dst): (*point).Hash (group/edwards25519/point.go:325-334), RFC 9380 suite
 * edwards25519_XMD:SHA-512_ELL2_RO_ (hashToField :336-360, expandMessageXMD :362-430, Elligator 2, cofactor 8).
 * Equal-length messages packed back to back; dst is a HOST pointer of 1..255 bytes. */
int kyb_ed25519_hash(size_t n, const uint8_t *msgs, size_t msg_len, const uint8_t *dst, size_t dst_len, uint8_t *out);
int kyb_ed25519_hash_dev(size_t n, const void *d_msgs, size_t msg_len, const uint8_t *dst, size_t dst_len,
                         void *d_out, void *stream);

/* Introspection used by the tests: copy the device-built fixed-base table
 * (33 x 8 entries of (y+x, y-x, 2dxy), 10 int32 limbs each) to the host. */
int kyb_ed25519_debug_base_table(int32_t *out /* 33*136*32: (position, |digit|-1, 30 limbs + 2 pad) */);

/* --------------------------------------------------------------- BLS12-381
 * The three reference suites pairing/bls12381/{kilic,circl,gnark} are adapters over external
 * modules (go.mod:6-8); their MarshalBinary wire formats are identical and are what crosses
 * this boundary:
 *   scalars : 32-byte big-endian (mod.Int, group/mod/int.go:334-350; kilic/scalar.go), taken as
 *             plain 256-bit integers
 *   G1      : 48-byte ZCash compressed (kilic/g1.go:119-131)      G2 : 96-byte (kilic/g2.go)
 *   GT      : 576 bytes (kilic/gt.go:115-117); layout documented in DESIGN.md (parity unpinned
 *             by the reference)
 * UnmarshalBinary semantics: flag rules + on-curve + subgroup membership, exactly the cases of
 * pairing/bls12381/deserialization_tests (bls12381_test.go:74-186).  A rejected input gives
 * status KYB_ST_BAD_POINT / KYB_ST_NOT_IN_SUBGROUP and an all-zero output element.          */

/* out[i] = scalars[i] * points[i].  Replaces G1Elt.UnmarshalBinary + Mul + MarshalBinary
 * (pairing/bls12381/kilic/g1.go:110-131; circl/g1.go:89-96; gnark/g1.go:118-127). */
int kyb_bls12381_g1_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
/* same on G2 (kilic/g2.go; this is what suite.Point().Mul is for the *.adapter suites) */
int kyb_bls12381_g2_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
/* out[i] = scalars[i] * point: the loop of share.PriPoly.Commit (share/poly.go:143-149).  Batches of 2^17 scalars and
 * more -- and, through the host-buffer calls, batches of 64 and more over the suite's generator or over the base of
 * the previous call -- run through a table of the base's multiples (26 table additions per scalar, no doublings; same
 * bytes and statuses as the per-element calls; the same holds for the bn256 / bn254 entry points and for the `_dev`
 * calls with point_stride = 0). */
int kyb_bls12381_g1_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[48], uint8_t *out,
                                  uint8_t *status, uint32_t flags);
int kyb_bls12381_g2_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[96], uint8_t *out,
                                  uint8_t *status, uint32_t flags);
/* point_stride: the input element size (48 / 96, or 96 / 192 with KYB_F_UNCOMPRESSED) for per-element points,
 * 0 for one shared base point */
int kyb_bls12381_g1_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                            void *d_status, uint32_t flags, void *stream);
int kyb_bls12381_g2_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                            void *d_status, uint32_t flags, void *stream);

/* out[i] = a[i] + b[i]: G1Elt.Add / G2Elt.Add (kilic/g1.go:90-96, g2.go). */
int kyb_bls12381_g1_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
int kyb_bls12381_g2_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
/* the same on device pointers, enqueued on `stream` (kilic/g1.go:90-96): what the node-wide MSM's combine of the gathered partial
 * points uses, so that nothing of the exchange step touches the host */
int kyb_bls12381_g1_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);
int kyb_bls12381_g2_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);

/* Batch UnmarshalBinary: status[i] = what G1Elt / G2Elt.UnmarshalBinary would decide (kilic/g1.go:127-131, g2.go:
 * ZCash flag rules, x < p, on the curve, in the r-torsion subgroup -- the 34 fixtures of
 * pairing/bls12381/deserialization_tests), out[i] = the point re-encoded (48 / 96 B, or 96 / 192 B affine with
 * KYB_F_UNCOMPRESSED_OUT; zero bytes when rejected).  Input is compressed unless KYB_F_UNCOMPRESSED.  This is the
 * call that earns KYB_F_TRUSTED(i) | KYB_F_UNCOMPRESSED on later calls with the same points. */
int kyb_bls12381_g1_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bls12381_g2_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bls12381_g1_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);
int kyb_bls12381_g2_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);

/* gt[i] = e(g1[i], g2[i]).  Replaces Suite.Pair (pairing/pairing.go:12; kilic/suite.go:70-75). */
int kyb_bls12381_pair(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, uint32_t flags);
int kyb_bls12381_pair_dev(size_t n, const void *d_g1, const void *d_g2, void *d_gt, void *d_status, uint32_t flags, void *stream);
/* out[i] = hash_to_curve(msgs[i], dst) on G1 / G2 (RFC 9380 BLS12381G1/G2_XMD:SHA-256_SSWU_RO_).  Replaces
 * G1Elt.Hash / G2Elt.Hash (pairing/bls12381/kilic/g1.go:161-170, g2.go; default DSTs kilic/g1.go:17, g2.go:18), the
 * step before the pairing check in sign/bls Verify (bls.go:87-88).  The n messages have the same length msg_len
 * and are packed back to back; dst is a HOST pointer (at most 255 bytes) in every variant. */
int kyb_bls12381_hash_g1(size_t n, const uint8_t *msgs, size_t msg_len, const uint8_t *dst, size_t dst_len,
                         uint8_t *out, uint8_t *status);
int kyb_bls12381_hash_g2(size_t n, const uint8_t *msgs, size_t msg_len, const uint8_t *dst, size_t dst_len,
                         uint8_t *out, uint8_t *status);
int kyb_bls12381_hash_g1_dev(size_t n, const void *d_msgs, size_t msg_len, const uint8_t *dst, size_t dst_len,
                             void *d_out, void *d_status, void *stream);
int kyb_bls12381_hash_g2_dev(size_t n, const void *d_msgs, size_t msg_len, const uint8_t *dst, size_t dst_len,
                             void *d_out, void *d_status, void *stream);
/* ok[i] = bls.Verify(pubkeys[i], msgs[i], sigs[i]) for the scheme with signatures on G1 and keys on G2
 * (sign/bls/bls.go:82-96 with the pairing closure of bls.go:36-38): e(H(msg), X) == e(sig, G2.Base()), hashing,
 * both UnmarshalBinary checks, two Miller loops and one final exponentiation fused per lane (the hashed point is
 * never encoded / re-decoded).  Equal-length messages packed back to back; dst is a HOST pointer.  status[i] != 0
 * (and ok[i] = 0) where the key or the signature does not unmarshal -- the reference returns an error there. */
int kyb_bls12381_verify_g1(size_t n, const uint8_t *pubkeys, const uint8_t *msgs, size_t msg_len, const uint8_t *dst,
                           size_t dst_len, const uint8_t *sigs, uint8_t *ok, uint8_t *status, uint32_t flags);
int kyb_bls12381_verify_g1_dev(size_t n, const void *d_pubkeys, const void *d_msgs, size_t msg_len,
                               const uint8_t *dst, size_t dst_len, const void *d_sigs, void *d_ok, void *d_status,
                               uint32_t flags, void *stream);
/* ok[i] = bls.Verify(pubkey, msgs[i], sigs[i]) for ONE public key (96 B, or 192 B with KYB_F_UNCOMPRESSED; a HOST
 * pointer in the first variant, a device pointer in `_dev`): sign/bls/bls.go:82-96 called in a loop with the same X --
 * a drand chain's beacons, the partial signatures of one tbls participant (sign/tbls/tbls.go:100-107).  With both G2
 * operands the same for every element, both Miller loops read their lines from tables (the generator's is a constant of
 * the program; the key's is built on the device the first time a key is seen and kept with those of the last eight keys
 * of the stream -- looked up on the device -- so that a verifier alternating between a few keys pays the walk once per key).  The key is checked as UnmarshalBinary would (KYB_F_TRUSTED(0) vouches for it);
 * a key that fails gives every element its status and ok = 0; status precedence per element: key, then signature. */
int kyb_bls12381_verify_g1_same_key(size_t n, const uint8_t *pubkey, const uint8_t *msgs, size_t msg_len,
                                    const uint8_t *dst, size_t dst_len, const uint8_t *sigs, uint8_t *ok,
                                    uint8_t *status, uint32_t flags);
int kyb_bls12381_verify_g1_same_key_dev(size_t n, const void *d_pubkey, const void *d_msgs, size_t msg_len,
                                        const uint8_t *dst, size_t dst_len, const void *d_sigs, void *d_ok,
                                        void *d_status, uint32_t flags, void *stream);
/* test hook: out3 = {cache hits, table builds, next slot} of `stream`'s key cache behind kyb_bls12381_verify_g1_same_key
 * (sign/bls/bls.go:82-96 over a committee of keys); synchronises the stream */
int kyb_bls12381_debug_vkey_stats(void *stream, uint32_t *out3);
/* ok[i] = bls.Verify(pubkeys[i], msg, sigs[i]) for ONE message: the verification loop of tbls.Recover
 * (sign/tbls/tbls.go:118-131 -- every partial signature of a round is over the same msg, each under its own public
 * share public.Eval(idx).V, share/poly.go:340-348).  H(msg) is computed once per call instead of once per element;
 * everything else is kyb_bls12381_verify_g1 (same flags, same status precedence: key, then signature).  `msg` is a HOST
 * pointer in the first variant, a device pointer in `_dev`; msg_len may be 0. */
int kyb_bls12381_verify_g1_same_msg(size_t n, const uint8_t *pubkeys, const uint8_t *msg, size_t msg_len,
                                    const uint8_t *dst, size_t dst_len, const uint8_t *sigs, uint8_t *ok,
                                    uint8_t *status, uint32_t flags);
int kyb_bls12381_verify_g1_same_msg_dev(size_t n, const void *d_pubkeys, const void *d_msg, size_t msg_len,
                                        const uint8_t *dst, size_t dst_len, const void *d_sigs, void *d_ok,
                                        void *d_status, uint32_t flags, void *stream);
/* The same for the scheme with signatures on G2 and keys on G1 (NewSchemeOnG2, sign/bls/bls.go:48-58:
 * ValidatePairing(G1.Base(), sig, X, H(msg)) with H = hash_to_curve on G2): pubkeys 48 B, sigs 96 B. */
int kyb_bls12381_verify_g2(size_t n, const uint8_t *pubkeys, const uint8_t *msgs, size_t msg_len, const uint8_t *dst,
                           size_t dst_len, const uint8_t *sigs, uint8_t *ok, uint8_t *status, uint32_t flags);
int kyb_bls12381_verify_g2_dev(size_t n, const void *d_pubkeys, const void *d_msgs, size_t msg_len,
                               const uint8_t *dst, size_t dst_len, const void *d_sigs, void *d_ok, void *d_status,
                               uint32_t flags, void *stream);
/* out[i] = gt[i] ^ scalars[i].  Replaces GTElt.Mul (kilic/gt.go:79-84 -> GT.Exp); inputs are checked
 * like GT.FromBytes (coefficients < p, order-r subgroup). */
int kyb_bls12381_gt_mul(size_t n, const uint8_t *scalars, const uint8_t *gt, uint8_t *out, uint8_t *status);
int kyb_bls12381_gt_mul_dev(size_t n, const void *d_scalars, const void *d_gt, void *d_out, void *d_status,
                            void *stream);
/* ok[i] = (e(p1[i], p2[i]) == e(inv1[i], inv2[i])).  Replaces Suite.ValidatePairing
 * (pairing/pairing.go:13-15; kilic/suite.go:57-68), the core of sign/bls Verify (bls.go:82-96). */
int kyb_bls12381_pair_check(size_t n, const uint8_t *p1, const uint8_t *p2, const uint8_t *inv1, const uint8_t *inv2,
                            uint8_t *ok, uint8_t *status, uint32_t flags);
int kyb_bls12381_pair_check_dev(size_t n, const void *d_p1, const void *d_p2, const void *d_inv1, const void *d_inv2,
                                void *d_ok, void *d_status, uint32_t flags, void *stream);

/* ------------------------------------------------------------------ bn256
 * pairing/bn256 (dclxvi parameters; arithmetic in-tree).  Wire formats (point.go):
 *   scalars : 32-byte big-endian (mod.Int), taken as plain 256-bit integers (curve.go:189-203)
 *   G1      : 64 bytes x || y, big-endian, infinity = 64 zero bytes (point.go:170-238)
 *   G2      : 128 bytes x.x || x.y || y.x || y.y with gfP2{x, y} = x i + y (point.go:423-499)
 *   GT      : 384 bytes, 12 x 32, order x.x.x ... y.z.y (point.go:630-662)
 * UnmarshalBinary semantics kept: coordinates are reduced mod p (not rejected), (0,0) is
 * infinity, on-curve check only -- G2 inputs outside the order-n subgroup are accepted and
 * processed like the reference does.  status: KYB_ST_BAD_POINT, output zeroed.
 * flags: KYB_F_TRUSTED(i) on a G2 operand is the caller's word that the point lies in the order-n subgroup (which this
 * suite's UnmarshalBinary never checks): g2_mul then walks a GLS decomposition and pair_check (below) the product form;
 * a vouched-for point outside the subgroup gives a result the reference could not.  Otherwise flags have no effect. */
int kyb_bn256_g1_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn256_g2_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn256_g1_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[64], uint8_t *out,
                               uint8_t *status, uint32_t flags);
int kyb_bn256_g2_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[128], uint8_t *out,
                               uint8_t *status, uint32_t flags);
int kyb_bn256_g1_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                         void *d_status, uint32_t flags, void *stream);
int kyb_bn256_g2_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                         void *d_status, uint32_t flags, void *stream);
/* out[i] = a[i] + b[i]: pointG1.Add / pointG2.Add (pairing/bn256/point.go:130-140, 381-391 -> curve.go:69). */
int kyb_bn256_g1_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
int kyb_bn256_g2_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
/* the same on device pointers, enqueued on `stream` (pairing/bn256/point.go:130-140): what the node-wide MSM's combine of the gathered partial
 * points uses, so that nothing of the exchange step touches the host */
int kyb_bn256_g1_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);
int kyb_bn256_g2_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);
/* Batch UnmarshalBinary: pointG1 / pointG2.UnmarshalBinary (pairing/bn256/point.go:206-238, 466-499): on the curve
 * (G2: on the twist, NO subgroup check, as the reference), 64 / 128 zero bytes = infinity; out[i] = MarshalBinary of
 * the accepted point (zero bytes when rejected).  flags are accepted and ignored. */
int kyb_bn256_g1_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn256_g2_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn256_g1_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);
int kyb_bn256_g2_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);

/* gt[i] = e(g1[i], g2[i]): Suite.Pair (pairing/bn256/suite.go:97-103 -> optate.go:266-274). */
int kyb_bn256_pair(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, uint32_t flags);
int kyb_bn256_pair_dev(size_t n, const void *d_g1, const void *d_g2, void *d_gt, void *d_status, uint32_t flags, void *stream);
/* out[i] = Hash(msgs[i]) on G1: pointG1.Hash -> hashToPoint (pairing/bn256/point.go:261-313), SHA-256 then
 * try-and-increment; the step before the pairing check in sign/bls Verify (bls.go:87-88).  All n messages
 * have the same length msg_len and are packed back to back (hash variable-length inputs down first). */
int kyb_bn256_hash_g1(size_t n, const uint8_t *msgs, size_t msg_len, uint8_t *out, uint8_t *status);
int kyb_bn256_hash_g1_dev(size_t n, const void *d_msgs, size_t msg_len, void *d_out, void *d_status, void *stream);
/* out[i] = HashG1(msgs[i], dst): the package-level hash of pairing/bn256/hash.go:10-110 -- hashToBase (gfp.go:46-68:
 * 48 bytes of HKDF-SHA-256 with secret = msg, salt = dst, info = "H2C" 0x00 0x01, reduced mod p) then the
 * Shallue-van de Woestijne map in the reference's arrangement (x1, x2, x3 tried in this order with legendre == 1,
 * y = the power (p + 1) / 4 with sign0(t)'s sign; s = sqrt(-3) of constants.go:105).  dst may be NULL / empty
 * (hash_test.go:11-20 passes nil), at most 255 bytes.  Pinned by the 11 outputs of hash_test.go:45-57. */
int kyb_bn256_hash_g1_svdw(size_t n, const uint8_t *msgs, size_t msg_len, const uint8_t *dst, size_t dst_len, uint8_t *out,
                           uint8_t *status);
int kyb_bn256_hash_g1_svdw_dev(size_t n, const void *d_msgs, size_t msg_len, const uint8_t *dst, size_t dst_len, void *d_out,
                               void *d_status, void *stream);
/* out[i] = gt[i] ^ scalars[i]: pointGT.Mul (pairing/bn256/point.go:613-628 -> gfP12.Exp gfp12.go:177);
 * like pointGT.UnmarshalBinary (point.go:664-716) coefficients are reduced mod p and nothing is rejected. */
int kyb_bn256_gt_mul(size_t n, const uint8_t *scalars, const uint8_t *gt, uint8_t *out, uint8_t *status);
int kyb_bn256_gt_mul_dev(size_t n, const void *d_scalars, const void *d_gt, void *d_out, void *d_status, void *stream);
/* ok[i] = Pair(p1, p2).Equal(Pair(inv1, inv2)): Suite.ValidatePairing (suite.go:105-107): two whole pairings and a
 * comparison, like the reference.  With KYB_F_TRUSTED(1) | KYB_F_TRUSTED(3) -- the caller vouches that both G2
 * operands lie in the order-n subgroup, which bn256's own UnmarshalBinary never checks -- the same predicate is
 * evaluated as e(p1, p2) e(-inv1, inv2) == 1 with one final exponentiation (1.6x the rate). */
int kyb_bn256_pair_check(size_t n, const uint8_t *p1, const uint8_t *p2, const uint8_t *inv1, const uint8_t *inv2,
                         uint8_t *ok, uint8_t *status, uint32_t flags);
int kyb_bn256_pair_check_dev(size_t n, const void *d_p1, const void *d_p2, const void *d_inv1, const void *d_inv2,
                             void *d_ok, void *d_status, uint32_t flags, void *stream);

/* ------------------------------------------------------------------ bn254
 * pairing/bn254 (Ethereum's alt_bn128; the bn256 package over other constants, xi = 9 + i).  Wire formats and entry
 * points as bn256 (scalars 32-byte big-endian, G1 64, G2 128, GT 384 bytes; point.go:127-200, 431-520, 617-735), with
 * this suite's stricter UnmarshalBinary kept:
 *   - a coordinate >= p is an error (gfp.go:101-118), not reduced                      -> KYB_ST_BAD_POINT
 *   - G2 points must lie in the order-n subgroup (twist.go:47-66: [Order]Q = infinity)  -> KYB_ST_NOT_IN_SUBGROUP
 *     KYB_F_TRUSTED(i) on a G2 operand skips that check (the caller unmarshalled the point before), as on BLS12-381.
 *   - GT coefficients >= p are rejected by gt_mul (point.go:662-735).
 * pointG1.Mul walks a GLV lattice decomposition (curve.go:196-222); the multiple is the same group element.
 * ValidatePairing = two pairings + Equal (suite.go:134-140).  No Hash on G2 (the reference has none). */
int kyb_bn254_g1_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn254_g2_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn254_g1_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[64], uint8_t *out,
                               uint8_t *status, uint32_t flags);
int kyb_bn254_g2_mul_same_base(size_t n, const uint8_t *scalars, const uint8_t point[128], uint8_t *out,
                               uint8_t *status, uint32_t flags);
int kyb_bn254_g1_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                         void *d_status, uint32_t flags, void *stream);
int kyb_bn254_g2_mul_dev(size_t n, const void *d_scalars, const void *d_points, size_t point_stride, void *d_out,
                         void *d_status, uint32_t flags, void *stream);
int kyb_bn254_g1_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
int kyb_bn254_g2_add(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *status);
/* the same on device pointers, enqueued on `stream` (pairing/bn254/point.go:92-101): what the node-wide MSM's combine of the gathered partial
 * points uses, so that nothing of the exchange step touches the host */
int kyb_bn254_g1_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);
int kyb_bn254_g2_add_dev(size_t n, const void *d_a, const void *d_b, void *d_out, void *d_status, void *stream);
int kyb_bn254_g1_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn254_g2_unmarshal(size_t n, const uint8_t *points, uint8_t *out, uint8_t *status, uint32_t flags);
int kyb_bn254_g1_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);
int kyb_bn254_g2_unmarshal_dev(size_t n, const void *d_points, void *d_out, void *d_status, uint32_t flags, void *stream);
int kyb_bn254_pair(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, uint32_t flags);
int kyb_bn254_pair_dev(size_t n, const void *d_g1, const void *d_g2, void *d_gt, void *d_status, uint32_t flags, void *stream);
int kyb_bn254_gt_mul(size_t n, const uint8_t *scalars, const uint8_t *gt, uint8_t *out, uint8_t *status);
int kyb_bn254_gt_mul_dev(size_t n, const void *d_scalars, const void *d_gt, void *d_out, void *d_status, void *stream);
int kyb_bn254_pair_check(size_t n, const uint8_t *p1, const uint8_t *p2, const uint8_t *inv1, const uint8_t *inv2,
                         uint8_t *ok, uint8_t *status, uint32_t flags);
int kyb_bn254_pair_check_dev(size_t n, const void *d_p1, const void *d_p2, const void *d_inv1, const void *d_inv2,
                             void *d_ok, void *d_status, uint32_t flags, void *stream);
/* out[i] = Hash(msgs[i]) on G1: pointG1.Hash -> hashToPoint (point.go:207-285): RFC 9380 hash_to_curve with
 * expand_message_xmd over legacy Keccak-256, the Shallue-van de Woestijne map (constants.go:72-84), no cofactor.
 * dst = the suite's domain separation tag (suite.go:42-44: "BN254G1_XMD:KECCAK-256_SVDW_RO_" unless SetDomainG1), at
 * most 255 bytes.  All n messages have the same length msg_len and are packed back to back. */
int kyb_bn254_hash_g1(size_t n, const uint8_t *msgs, size_t msg_len, const uint8_t *dst, size_t dst_len, uint8_t *out,
                      uint8_t *status);
int kyb_bn254_hash_g1_dev(size_t n, const void *d_msgs, size_t msg_len, const uint8_t *dst, size_t dst_len, void *d_out,
                          void *d_status, void *stream);

/* ------------------------------------------------------- multi-scalar multiplication
 * out = sum_i scalars[i] * points[i]  (one point).  The reference has no MSM function: its
 * MSM-shaped call sites do N x (Mul + Add) sequentially -- share.PubPoly.Eval (share/poly.go:340-348),
 * share.RecoverCommit (share/poly.go:449-476), bdn.AggregateSignatures / AggregatePublicKeys
 * (sign/bdn/bdn.go:126-181, mask.go:57-61).  Encodings are canonical, so the Pippenger result is
 * byte-identical to that sequential sum -- for every 32-byte scalar: an Ed25519 scalar that the
 * reference's radix-16 recoding mangles (top digit above 8, see kyb_ed25519_mul) counts as the integer
 * the reference's Mul multiplies by.  status[i] reports undecodable inputs; if any input is
 * rejected the output is all-zero bytes.  n == 0 yields the encoding of the identity.
 * The _dev variants work in a grow-only workspace per (device, stream).                            */
int kyb_ed25519_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[32], uint8_t *status);
/* the same with flags (KYB_F_SCALAR_BITS) */
int kyb_ed25519_msm_flags(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[32], uint8_t *status,
                          uint32_t flags);
int kyb_ed25519_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                        void *stream);
int kyb_bls12381_g1_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[48], uint8_t *status, uint32_t flags);
int kyb_bls12381_g2_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[96], uint8_t *status, uint32_t flags);
int kyb_bls12381_g1_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                            uint32_t flags, void *stream);
int kyb_bls12381_g2_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                            uint32_t flags, void *stream);
int kyb_bn256_g1_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[64], uint8_t *status, uint32_t flags);
int kyb_bn256_g2_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[128], uint8_t *status, uint32_t flags);
int kyb_bn256_g1_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                         uint32_t flags, void *stream);
int kyb_bn256_g2_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                         uint32_t flags, void *stream);

/* --------------------------------------------------------- batched public-polynomial evaluation
 * out[i] = sum_j commits[j] * (idx[i] + 1)^j  (n points).  share.PubPoly.Eval (share/poly.go:340-348: xi = 1 + i,
 * Horner from the top coefficient with t x (Mul + Add)) for many indices in one launch -- the per-participant loop of
 * PubPoly.Shares / Check and of the DKG / VSS share verification (share/dkg/pedersen/dkg.go:294, 489-490, 825-826).
 * commits: t encoded points (coefficient 0 first); status[j] reports an undecodable commitment, and then every
 * output is all-zero bytes.  t == 0 yields identities.  flags as for the MSM (KYB_F_TRUSTED(0) = the commitments).  */
int kyb_ed25519_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                          uint8_t *status);
int kyb_ed25519_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                              void *d_status, void *stream);
int kyb_bls12381_g1_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                              uint8_t *status, uint32_t flags);
int kyb_bls12381_g2_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                              uint8_t *status, uint32_t flags);
int kyb_bls12381_g1_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                                  void *d_status, uint32_t flags, void *stream);
int kyb_bls12381_g2_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                                  void *d_status, uint32_t flags, void *stream);
int kyb_bn256_g1_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                           uint8_t *status, uint32_t flags);
int kyb_bn256_g2_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                           uint8_t *status, uint32_t flags);
int kyb_bn256_g1_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                               void *d_status, uint32_t flags, void *stream);
int kyb_bn256_g2_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                               void *d_status, uint32_t flags, void *stream);
/* --------------------------------------------------------- batched private-polynomial evaluation (scalar field)
 * out[i] = sum_j coeffs[j] * (idx[i] + 1)^j mod q  (n scalars of 32 bytes, q = the group order).  share.PriPoly.Eval
 * (share/poly.go:85-93: xi = 1 + i, Horner from the top coefficient with t x (Mul + Add) of group/mod.Int) for many
 * indices in one launch -- the loop of PriPoly.Shares (share/poly.go:96-102), which every dealer of share/vss and
 * share/dkg runs once per participant.  coeffs: t scalars in the suite's scalar encoding (Ed25519: 32 bytes
 * little-endian, group/edwards25519/scalar.go; pairing suites: mod.Int's 32 bytes big-endian, group/mod/int.go);
 * any 32-byte string is taken modulo q; outputs are canonical.  t == 0 yields zeros.                              */
int kyb_ed25519_scalar_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *coeffs, uint8_t *out);
int kyb_ed25519_scalar_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_coeffs, void *d_out, void *stream);
int kyb_bls12381_scalar_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *coeffs, uint8_t *out);
int kyb_bls12381_scalar_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_coeffs, void *d_out, void *stream);
int kyb_bn256_scalar_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *coeffs, uint8_t *out);
int kyb_bn256_scalar_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_coeffs, void *d_out, void *stream);
int kyb_bn254_scalar_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *coeffs, uint8_t *out);
int kyb_bn254_scalar_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_coeffs, void *d_out, void *stream);
/* pairing/bn254: the same eight entry points */
int kyb_bn254_g1_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[64], uint8_t *status, uint32_t flags);
int kyb_bn254_g2_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[128], uint8_t *status, uint32_t flags);
int kyb_bn254_g1_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                         uint32_t flags, void *stream);
int kyb_bn254_g2_msm_dev(size_t n, const void *d_scalars, const void *d_points, void *d_out, void *d_status,
                         uint32_t flags, void *stream);
int kyb_bn254_g1_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                           uint8_t *status, uint32_t flags);
int kyb_bn254_g2_poly_eval(size_t n, const uint32_t *idx, size_t t, const uint8_t *commits, uint8_t *out,
                           uint8_t *status, uint32_t flags);
int kyb_bn254_g1_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                               void *d_status, uint32_t flags, void *stream);
int kyb_bn254_g2_poly_eval_dev(size_t n, const void *d_idx, size_t t, const void *d_commits, void *d_out,
                               void *d_status, uint32_t flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif
