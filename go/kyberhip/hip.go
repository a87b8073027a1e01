// Package kyberhip binds libkyberhip.so (include/kyber_hip.h), the MI355X batch engine for kyber's
// Point.Mul / multi-scalar-mul / pairing hot path.
//
// NOT COMPILED in the repository that ships it (that image has no Go toolchain): it is the binding a
// kyber maintainer would drop into the module, kept next to the C ABI so the two stay in step.  Build
// with: go build -tags hip ./...   (CGO_ENABLED=1, libkyberhip.so built by __graft_entry__.build()).
//
// Every function takes the concatenated MarshalBinary encodings the reference produces and returns the
// concatenated encodings of the results plus one status byte per element (0 = ok, 1 = the reference's
// UnmarshalBinary would have failed, 2 = BLS12-381 point outside the prime-order subgroup).
//
//go:build hip

package kyberhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../kyber_amd/lib -lkyberhip -Wl,-rpath,${SRCDIR}/../../kyber_amd/lib
#include "kyber_hip.h"
*/
import "C"

import (
	"errors"
	"unsafe"
)

// Flags of the pairing-suite calls (kyber_hip.h).
const (
	Vartime         = uint32(C.KYB_F_VARTIME)          // Ed25519: geScalarMultVartime semantics
	Uncompressed    = uint32(C.KYB_F_UNCOMPRESSED)     // BLS12-381 inputs in the 96 / 192-byte uncompressed form
	UncompressedOut = uint32(C.KYB_F_UNCOMPRESSED_OUT) // BLS12-381 mul outputs too
	TrustedAll      = uint32(C.KYB_F_TRUSTED_ALL)
)

// Trusted marks point argument arg (0-based) as the encoding of an already-validated kyber.Point.
func Trusted(arg uint) uint32 { return 0x100 << arg }

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func check(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return errors.New("kyberhip: " + C.GoString(C.kyb_last_error()))
}

// Init creates the context of the calling thread's current HIP device (tables, pools); optional.
func Init() error { return check(C.kyb_init()) }

// StreamRelease frees the per-stream device workspaces tied to a HIP stream handle; call it before destroying a
// stream that was passed to the *_dev entry points.
func StreamRelease(stream unsafe.Pointer) error { return check(C.kyb_stream_release(stream)) }

// ---------------------------------------------------------------- Ed25519 (32-byte LE scalars, 32-byte points)

func Ed25519MulBase(scalars []byte, flags uint32) (out []byte, err error) {
	n := len(scalars) / 32
	out = make([]byte, 32*n)
	err = check(C.kyb_ed25519_mul_base(C.size_t(n), ptr(scalars), ptr(out), C.uint32_t(flags)))
	return
}

func Ed25519Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 32*n), make([]byte, n)
	err = check(C.kyb_ed25519_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Ed25519MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 32*n), make([]byte, n)
	err = check(C.kyb_ed25519_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status),
		C.uint32_t(flags)))
	return
}

func Ed25519MSM(scalars, points []byte) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 32), make([]byte, n)
	err = check(C.kyb_ed25519_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status)))
	return
}

func Ed25519Add(a, b []byte) (out, status []byte, err error) {
	n := len(a) / 32
	out, status = make([]byte, 32*n), make([]byte, n)
	err = check(C.kyb_ed25519_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)))
	return
}

// Ed25519Unmarshal: batch (*point).UnmarshalBinary; status[i] != 0 where the reference returns an error, out[i] is
// the canonical re-encoding.
func Ed25519Unmarshal(points []byte) (out, status []byte, err error) {
	n := len(points) / 32
	out, status = make([]byte, 32*n), make([]byte, n)
	err = check(C.kyb_ed25519_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status)))
	return
}

// Ed25519PolyEval: out[i] = sum_j commits[j] * (idx[i] + 1)^j  (share.PubPoly.Eval for many indices).
func Ed25519PolyEval(idx []uint32, commits []byte) (out, status []byte, err error) {
	n, t := len(idx), len(commits)/32
	out, status = make([]byte, 32*n), make([]byte, t)
	var ip *C.uint32_t
	if n > 0 {
		ip = (*C.uint32_t)(unsafe.Pointer(&idx[0]))
	}
	err = check(C.kyb_ed25519_poly_eval(C.size_t(n), ip, C.size_t(t), ptr(commits), ptr(out), ptr(status)))
	return
}

// Ed25519Hash: n equal-length messages packed back to back -> n points (RFC 9380 edwards25519_XMD:SHA-512_ELL2_RO_).
func Ed25519Hash(msgs []byte, msgLen int, dst []byte) (out []byte, err error) {
	n := 0
	if msgLen > 0 {
		n = len(msgs) / msgLen
	}
	out = make([]byte, 32*n)
	err = check(C.kyb_ed25519_hash(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out)))
	return
}

// ---------------------------------------------------------------- BLS12-381 (32-byte BE scalars; G1 48 B, G2 96 B, GT 576 B)

func g1in(flags uint32) int {
	if flags&Uncompressed != 0 {
		return 96
	}
	return 48
}
func g2in(flags uint32) int { return 2 * g1in(flags) }
func g1out(flags uint32) int {
	if flags&UncompressedOut != 0 {
		return 96
	}
	return 48
}

func Bls12381G1Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g1_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Bls12381G2Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g2_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Bls12381G1MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g1_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status),
		C.uint32_t(flags)))
	return
}

func Bls12381G2MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g2_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status),
		C.uint32_t(flags)))
	return
}

func Bls12381G1MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 48), make([]byte, n)
	err = check(C.kyb_bls12381_g1_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Bls12381G2MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 96), make([]byte, n)
	err = check(C.kyb_bls12381_g2_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

// Bls12381G1PolyEval: share.PubPoly.Eval over G1 for many indices in one launch.
func Bls12381G1PolyEval(idx []uint32, commits []byte, flags uint32) (out, status []byte, err error) {
	n, t := len(idx), len(commits)/g1in(flags)
	out, status = make([]byte, 48*n), make([]byte, t)
	var ip *C.uint32_t
	if n > 0 {
		ip = (*C.uint32_t)(unsafe.Pointer(&idx[0]))
	}
	err = check(C.kyb_bls12381_g1_poly_eval(C.size_t(n), ip, C.size_t(t), ptr(commits), ptr(out), ptr(status),
		C.uint32_t(flags)))
	return
}

// Bls12381G1Unmarshal / G2Unmarshal: batch UnmarshalBinary (ZCash rules + subgroup check).  With UncompressedOut the
// result is the affine form later calls accept under Uncompressed | Trusted(i).
func Bls12381G1Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n := len(points) / g1in(flags)
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g1_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Bls12381G2Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n := len(points) / g2in(flags)
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = check(C.kyb_bls12381_g2_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)))
	return
}

func Bls12381Pair(g1, g2 []byte, flags uint32) (gt, status []byte, err error) {
	n := len(g1) / g1in(flags)
	gt, status = make([]byte, 576*n), make([]byte, n)
	err = check(C.kyb_bls12381_pair(C.size_t(n), ptr(g1), ptr(g2), ptr(gt), ptr(status), C.uint32_t(flags)))
	return
}

// Bls12381ValidatePairing: ok[i] = e(p1[i], p2[i]) == e(inv1[i], inv2[i]); Trusted(0..3) = p1, p2, inv1, inv2.
func Bls12381ValidatePairing(p1, p2, inv1, inv2 []byte, flags uint32) (ok, status []byte, err error) {
	n := len(p1) / g1in(flags)
	ok, status = make([]byte, n), make([]byte, n)
	err = check(C.kyb_bls12381_pair_check(C.size_t(n), ptr(p1), ptr(p2), ptr(inv1), ptr(inv2), ptr(ok), ptr(status),
		C.uint32_t(flags)))
	return
}

func Bls12381HashG1(msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	n := len(msgs) / msgLen
	out, status = make([]byte, 48*n), make([]byte, n)
	err = check(C.kyb_bls12381_hash_g1(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out),
		ptr(status)))
	return
}

func Bls12381HashG2(msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	n := len(msgs) / msgLen
	out, status = make([]byte, 96*n), make([]byte, n)
	err = check(C.kyb_bls12381_hash_g2(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out),
		ptr(status)))
	return
}

// Bls12381VerifyG1: n x sign/bls Verify (signatures on G1, keys on G2) in one fused kernel; Trusted(0) = keys.
func Bls12381VerifyG1(pubkeys, msgs []byte, msgLen int, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n := len(sigs) / g1in(flags)
	ok, status = make([]byte, n), make([]byte, n)
	err = check(C.kyb_bls12381_verify_g1(C.size_t(n), ptr(pubkeys), ptr(msgs), C.size_t(msgLen), ptr(dst),
		C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags)))
	return
}

// Bls12381VerifyG2: the same for signatures on G2 and keys on G1 (NewSchemeOnG2); Trusted(0) = keys.
func Bls12381VerifyG2(pubkeys, msgs []byte, msgLen int, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n := len(sigs) / g2in(flags)
	ok, status = make([]byte, n), make([]byte, n)
	err = check(C.kyb_bls12381_verify_g2(C.size_t(n), ptr(pubkeys), ptr(msgs), C.size_t(msgLen), ptr(dst),
		C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags)))
	return
}

// ---------------------------------------------------------------- bn256 (32-byte BE scalars; G1 64 B, G2 128 B, GT 384 B)

func Bn256G1Mul(scalars, points []byte) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 64*n), make([]byte, n)
	err = check(C.kyb_bn256_g1_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0))
	return
}

func Bn256G2Mul(scalars, points []byte) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 128*n), make([]byte, n)
	err = check(C.kyb_bn256_g2_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0))
	return
}

func Bn256G1MSM(scalars, points []byte) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 64), make([]byte, n)
	err = check(C.kyb_bn256_g1_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0))
	return
}

func Bn256G2MSM(scalars, points []byte) (out, status []byte, err error) {
	n := len(scalars) / 32
	out, status = make([]byte, 128), make([]byte, n)
	err = check(C.kyb_bn256_g2_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0))
	return
}

// Bn256G1Unmarshal / G2Unmarshal: batch UnmarshalBinary (on-curve check only, as the reference).
func Bn256G1Unmarshal(points []byte) (out, status []byte, err error) {
	n := len(points) / 64
	out, status = make([]byte, 64*n), make([]byte, n)
	err = check(C.kyb_bn256_g1_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), 0))
	return
}

func Bn256G2Unmarshal(points []byte) (out, status []byte, err error) {
	n := len(points) / 128
	out, status = make([]byte, 128*n), make([]byte, n)
	err = check(C.kyb_bn256_g2_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), 0))
	return
}

func Bn256Pair(g1, g2 []byte) (gt, status []byte, err error) {
	n := len(g1) / 64
	gt, status = make([]byte, 384*n), make([]byte, n)
	err = check(C.kyb_bn256_pair(C.size_t(n), ptr(g1), ptr(g2), ptr(gt), ptr(status), 0))
	return
}

func Bn256ValidatePairing(p1, p2, inv1, inv2 []byte) (ok, status []byte, err error) {
	n := len(p1) / 64
	ok, status = make([]byte, n), make([]byte, n)
	err = check(C.kyb_bn256_pair_check(C.size_t(n), ptr(p1), ptr(p2), ptr(inv1), ptr(inv2), ptr(ok), ptr(status), 0))
	return
}

func Bn256HashG1(msgs []byte, msgLen int) (out, status []byte, err error) {
	n := len(msgs) / msgLen
	out, status = make([]byte, 64*n), make([]byte, n)
	err = check(C.kyb_bn256_hash_g1(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(out), ptr(status)))
	return
}
