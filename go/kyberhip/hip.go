// Package kyberhip binds libkyberhip.so (include/kyber_hip.h), the MI355X batch engine for kyber's
// Point.Mul / multi-scalar-mul / pairing hot path.
//
// NOT COMPILED in the repository that ships it (that image has no Go toolchain): it is the binding a
// kyber maintainer drops into the module (go.dedis.ch/kyber/v4/hip), kept next to the C ABI so the two
// stay in step -- tests/test_cabi.py checks every C call here against the header (name and arity), and
// the suite package against the interfaces of group.go / pairing/pairing.go.
// Build with: go build -tags hip ./...   (CGO_ENABLED=1, libkyberhip.so built by __graft_entry__.build()).
//
// Every function takes the concatenated MarshalBinary encodings the reference produces and returns the
// concatenated encodings of the results plus one status byte per element (0 = ok, 1 = the reference's
// UnmarshalBinary would have failed, 2 = BLS12-381 point outside the prime-order subgroup).
//
// The engine is VARIABLE-TIME on every path (secret-indexed table loads, data-dependent inversion
// loops): suites built on it must not be registered as constant-time (suites.RequireConstantTime).
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
	"fmt"
	"runtime"
	"unsafe"
)

// Flags of the calls (kyber_hip.h).
const (
	Vartime         = uint32(C.KYB_F_VARTIME)          // Ed25519: geScalarMultVartime semantics
	Uniform         = uint32(C.KYB_F_UNIFORM)          // Ed25519: tables scanned, not indexed (the constant-time path's access pattern); exclusive with Vartime
	Uncompressed    = uint32(C.KYB_F_UNCOMPRESSED)     // BLS12-381 inputs in the 96 / 192-byte uncompressed form
	UncompressedOut = uint32(C.KYB_F_UNCOMPRESSED_OUT) // BLS12-381 mul outputs too
	TrustedAll      = uint32(C.KYB_F_TRUSTED_ALL)      // the four point arguments of a pairing check
)

// Trusted marks point argument arg (0-based) as the encoding of an already-validated kyber.Point.
func Trusted(arg uint) uint32 { return 0x100 << arg }

// ScalarBits tells an MSM that every scalar is below 2^bits (sign/bdn's 128-bit coefficients).
func ScalarBits(bits uint) uint32 { return uint32(bits) << 16 }

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// call runs one C entry point and fetches its error text on the SAME OS thread: kyb_last_error() is
// thread-local, and so is the HIP current device, while a goroutine may migrate between threads.
func call(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := f(); rc != 0 {
		return errors.New("kyberhip: " + C.GoString(C.kyb_last_error()))
	}
	return nil
}

// need rejects a slice whose length is not n elements of size bytes: the C side trusts its sizes.
func need(what string, b []byte, n, size int) error {
	if len(b) != n*size {
		return fmt.Errorf("kyberhip: %s holds %d bytes, want %d x %d", what, len(b), n, size)
	}
	return nil
}

func count(what string, b []byte, size int) (int, error) {
	if size <= 0 || len(b)%size != 0 {
		return 0, fmt.Errorf("kyberhip: %s holds %d bytes, not a multiple of %d", what, len(b), size)
	}
	return len(b) / size, nil
}

func firstErr(errs ...error) error {
	for _, e := range errs {
		if e != nil {
			return e
		}
	}
	return nil
}

// Init creates the context of the calling thread's current HIP device (tables, pools); optional.
func Init() error { return call(func() C.int { return C.kyb_init() }) }

// InitDevices makes every host-buffer batch call use devices 0 .. ndev-1 of the node: batches are cut into
// contiguous slices, one host thread and device context per slice; an MSM adds the per-device partial points.
func InitDevices(ndev int) error { return call(func() C.int { return C.kyb_init_devices(C.int(ndev)) }) }

// SetDevices selects the devices explicitly (nil: single-device behaviour).
func SetDevices(devices []int) error {
	d := make([]C.int, len(devices)+1)
	for i, v := range devices {
		d[i] = C.int(v)
	}
	return call(func() C.int { return C.kyb_set_devices(&d[0], C.int(len(devices))) })
}

// SetShardThreshold: host batches below minUnits stay on the caller's device.
func SetShardThreshold(minUnits int) error {
	return call(func() C.int { return C.kyb_set_shard_threshold(C.size_t(minUnits)) })
}

// DeviceCount returns the number of visible devices.
func DeviceCount() int { return int(C.kyb_device_count()) }

// StreamRelease frees the per-stream device workspaces tied to a HIP stream handle; call it before destroying a
// stream that was passed to the *_dev entry points.
func StreamRelease(stream unsafe.Pointer) error {
	return call(func() C.int { return C.kyb_stream_release(stream) })
}

// ---------------------------------------------------------------- Ed25519 (32-byte LE scalars, 32-byte points)

func Ed25519MulBase(scalars []byte, flags uint32) (out []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err != nil {
		return nil, err
	}
	out = make([]byte, 32*n)
	err = call(func() C.int { return C.kyb_ed25519_mul_base(C.size_t(n), ptr(scalars), ptr(out), C.uint32_t(flags)) })
	return
}

func Ed25519Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 32)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 32*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_ed25519_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Ed25519MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, 32)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 32*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_ed25519_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

// Ed25519MSM: sum_i scalars[i] * points[i]; flags: ScalarBits(b) or 0.
func Ed25519MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 32)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 32), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_ed25519_msm_flags(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

func Ed25519Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 32)
	if err = firstErr(err, need("b", b, n, 32)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 32*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_ed25519_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

// Ed25519Unmarshal: batch (*point).UnmarshalBinary; status[i] != 0 where the reference returns an error, out[i] is
// the canonical re-encoding.
func Ed25519Unmarshal(points []byte) (out, status []byte, err error) {
	n, err := count("points", points, 32)
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 32*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_ed25519_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status)) })
	return
}

func idxPtr(idx []uint32) *C.uint32_t {
	if len(idx) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&idx[0]))
}

// Ed25519PolyEval: out[i] = sum_j commits[j] * (idx[i] + 1)^j  (share.PubPoly.Eval for many indices).
func Ed25519PolyEval(idx []uint32, commits []byte) (out, status []byte, err error) {
	t, err := count("commits", commits, 32)
	if err != nil {
		return nil, nil, err
	}
	n := len(idx)
	out, status = make([]byte, 32*n), make([]byte, t+1)
	err = call(func() C.int {
		return C.kyb_ed25519_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(commits), ptr(out), ptr(status))
	})
	return out, status[:t], err
}

// ScalarPolyEval: out[i] = sum_j coeffs[j] * (idx[i] + 1)^j mod the group order -- share.PriPoly.Eval
// (share/poly.go:85-93) for many indices in one launch, i.e. PriPoly.Shares (share/poly.go:96-102).  coeffs and out are
// 32-byte scalars in the suite's own encoding (Ed25519 little-endian, mod.Int big-endian).
func ScalarPolyEval(suite string, idx []uint32, coeffs []byte) (out []byte, err error) {
	t, err := count("coeffs", coeffs, 32)
	if err != nil {
		return nil, err
	}
	switch suite {
	case "ed25519", "bls12381", "bn256", "bn254":
	default: // a mistyped suite name must not fall through to some other group order
		return nil, fmt.Errorf("kyberhip: ScalarPolyEval: unknown suite %q", suite)
	}
	n := len(idx)
	out = make([]byte, 32*n)
	err = call(func() C.int {
		switch suite {
		case "ed25519":
			return C.kyb_ed25519_scalar_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(coeffs), ptr(out))
		case "bls12381":
			return C.kyb_bls12381_scalar_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(coeffs), ptr(out))
		case "bn256":
			return C.kyb_bn256_scalar_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(coeffs), ptr(out))
		default: // "bn254": the names were checked above
			return C.kyb_bn254_scalar_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(coeffs), ptr(out))
		}
	})
	return out, err
}

// messages: n equal-length messages packed back to back (msgLen may be 0: then n must be given by the caller)
func messages(msgs []byte, msgLen, n int) error {
	if msgLen < 0 || len(msgs) != n*msgLen {
		return fmt.Errorf("kyberhip: %d message bytes, want %d x %d", len(msgs), n, msgLen)
	}
	return nil
}

// Ed25519Hash: n equal-length messages packed back to back -> n points (RFC 9380 edwards25519_XMD:SHA-512_ELL2_RO_).
func Ed25519Hash(n int, msgs []byte, msgLen int, dst []byte) (out []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, err
	}
	out = make([]byte, 32*n)
	err = call(func() C.int {
		return C.kyb_ed25519_hash(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out))
	})
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
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, g1in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g1_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G2Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, g2in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g2_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G1MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, g1in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g1_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G2MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, g2in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g2_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G1MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, g1in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 48), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bls12381_g1_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

func Bls12381G2MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, g2in(flags))); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 96), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bls12381_g2_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

// Bls12381G1PolyEval: share.PubPoly.Eval over G1 for many indices in one launch.
func Bls12381G1PolyEval(idx []uint32, commits []byte, flags uint32) (out, status []byte, err error) {
	t, err := count("commits", commits, g1in(flags))
	if err != nil {
		return nil, nil, err
	}
	n := len(idx)
	out, status = make([]byte, 48*n), make([]byte, t+1)
	err = call(func() C.int {
		return C.kyb_bls12381_g1_poly_eval(C.size_t(n), idxPtr(idx), C.size_t(t), ptr(commits), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:t], err
}

// Bls12381G1Unmarshal / G2Unmarshal: batch UnmarshalBinary (ZCash rules + subgroup check).  With UncompressedOut the
// result is the affine form later calls accept under Uncompressed | Trusted(i).
func Bls12381G1Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("points", points, g1in(flags))
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g1_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G2Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("points", points, g2in(flags))
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 2*g1out(flags)*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_g2_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bls12381G1Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 48)
	if err = firstErr(err, need("b", b, n, 48)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 48*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bls12381_g1_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bls12381G2Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 96)
	if err = firstErr(err, need("b", b, n, 96)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 96*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bls12381_g2_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bls12381Pair(g1, g2 []byte, flags uint32) (gt, status []byte, err error) {
	n, err := count("g1", g1, g1in(flags))
	if err = firstErr(err, need("g2", g2, n, g2in(flags))); err != nil {
		return nil, nil, err
	}
	gt, status = make([]byte, 576*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_pair(C.size_t(n), ptr(g1), ptr(g2), ptr(gt), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bls12381ValidatePairing: ok[i] = e(p1[i], p2[i]) == e(inv1[i], inv2[i]); Trusted(0..3) = p1, p2, inv1, inv2.
func Bls12381ValidatePairing(p1, p2, inv1, inv2 []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("p1", p1, g1in(flags))
	if err = firstErr(err, need("p2", p2, n, g2in(flags)), need("inv1", inv1, n, g1in(flags)), need("inv2", inv2, n, g2in(flags))); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_pair_check(C.size_t(n), ptr(p1), ptr(p2), ptr(inv1), ptr(inv2), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bls12381GTMul: out[i] = gt[i]^scalars[i]  (GTElt.Mul, kilic/gt.go:79-84).
func Bls12381GTMul(scalars, gt []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("gt", gt, n, 576)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 576*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bls12381_gt_mul(C.size_t(n), ptr(scalars), ptr(gt), ptr(out), ptr(status)) })
	return
}

func Bls12381HashG1(n int, msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 48*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_hash_g1(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out), ptr(status))
	})
	return
}

func Bls12381HashG2(n int, msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 96*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_hash_g2(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out), ptr(status))
	})
	return
}

// Bls12381VerifyG1SameKey: n x sign/bls Verify under ONE public key (a drand chain; sign/tbls/tbls.go:100-107): both
// Miller loops from line tables, the key's built once per key on the device.  Trusted(0) = the key.
func Bls12381VerifyG1SameKey(pubkey, msgs []byte, msgLen int, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("sigs", sigs, g1in(flags))
	if err = firstErr(err, need("pubkey", pubkey, 1, g2in(flags)), messages(msgs, msgLen, n)); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_verify_g1_same_key(C.size_t(n), ptr(pubkey), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bls12381VerifyG1SameMsg: ok[i] = bls.Verify(pubkeys[i], msg, sigs[i]) for ONE message -- the verification loop of
// tbls.Recover (sign/tbls/tbls.go:118-131); H(msg) is hashed once per call.
func Bls12381VerifyG1SameMsg(pubkeys, msg, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("sigs", sigs, g1in(flags))
	if err = firstErr(err, need("pubkeys", pubkeys, n, g2in(flags))); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_verify_g1_same_msg(C.size_t(n), ptr(pubkeys), ptr(msg), C.size_t(len(msg)), ptr(dst), C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bls12381VerifyG1: n x sign/bls Verify (signatures on G1, keys on G2), hash + checks + product of two Miller loops
// + final exponentiation on the device; Trusted(0) = keys.
func Bls12381VerifyG1(pubkeys, msgs []byte, msgLen int, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("sigs", sigs, g1in(flags))
	if err = firstErr(err, need("pubkeys", pubkeys, n, g2in(flags)), messages(msgs, msgLen, n)); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_verify_g1(C.size_t(n), ptr(pubkeys), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bls12381VerifyG2: the same for signatures on G2 and keys on G1 (NewSchemeOnG2); Trusted(0) = keys.
func Bls12381VerifyG2(pubkeys, msgs []byte, msgLen int, dst, sigs []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("sigs", sigs, g2in(flags))
	if err = firstErr(err, need("pubkeys", pubkeys, n, g1in(flags)), messages(msgs, msgLen, n)); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bls12381_verify_g2(C.size_t(n), ptr(pubkeys), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(sigs), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// ---------------------------------------------------------------- bn256 (32-byte BE scalars; G1 64 B, G2 128 B, GT 384 B)

func Bn256G1Mul(scalars, points []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g1_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0) })
	return
}

func Bn256G2Mul(scalars, points []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g2_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), 0) })
	return
}

func Bn256G1MulSameBase(scalars, point []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn256_g1_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), 0)
	})
	return
}

func Bn256G2MulSameBase(scalars, point []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn256_g2_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), 0)
	})
	return
}

func Bn256G1MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bn256_g1_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

func Bn256G2MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bn256_g2_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

// Bn256G1Unmarshal / G2Unmarshal: batch UnmarshalBinary (on-curve check only, as the reference).
func Bn256G1Unmarshal(points []byte) (out, status []byte, err error) {
	n, err := count("points", points, 64)
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g1_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), 0) })
	return
}

func Bn256G2Unmarshal(points []byte) (out, status []byte, err error) {
	n, err := count("points", points, 128)
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g2_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), 0) })
	return
}

func Bn256G1Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 64)
	if err = firstErr(err, need("b", b, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g1_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bn256G2Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 128)
	if err = firstErr(err, need("b", b, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_g2_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bn256Pair(g1, g2 []byte) (gt, status []byte, err error) {
	n, err := count("g1", g1, 64)
	if err = firstErr(err, need("g2", g2, n, 128)); err != nil {
		return nil, nil, err
	}
	gt, status = make([]byte, 384*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_pair(C.size_t(n), ptr(g1), ptr(g2), ptr(gt), ptr(status), 0) })
	return
}

func Bn256ValidatePairing(p1, p2, inv1, inv2 []byte) (ok, status []byte, err error) {
	n, err := count("p1", p1, 64)
	if err = firstErr(err, need("p2", p2, n, 128), need("inv1", inv1, n, 64), need("inv2", inv2, n, 128)); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn256_pair_check(C.size_t(n), ptr(p1), ptr(p2), ptr(inv1), ptr(inv2), ptr(ok), ptr(status), 0)
	})
	return
}

// Bn256GTMul: out[i] = gt[i]^scalars[i]  (pointGT.Mul, point.go:613 -> gfP12.Exp).
func Bn256GTMul(scalars, gt []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("gt", gt, n, 384)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 384*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_gt_mul(C.size_t(n), ptr(scalars), ptr(gt), ptr(out), ptr(status)) })
	return
}

func Bn256HashG1(n int, msgs []byte, msgLen int) (out, status []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn256_hash_g1(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(out), ptr(status)) })
	return
}

// ---------------------------------------------------------------- bn254 (alt_bn128; formats as bn256)
// pairing/bn254 rejects coordinates >= p and G2 points outside the subgroup; Trusted(i) on a G2 operand that was
// unmarshalled before skips the subgroup re-check (flags argument of every call that decodes points).

func Bn254G1Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g1_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)) })
	return
}

func Bn254G2Mul(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g2_mul(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)) })
	return
}

func Bn254G1MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn254_g1_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bn254G2MulSameBase(scalars, point []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("point", point, 1, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn254_g2_mul_same_base(C.size_t(n), ptr(scalars), ptr(point), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return
}

func Bn254G1MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bn254_g1_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

func Bn254G2MSM(scalars, points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("points", points, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128), make([]byte, n+1)
	err = call(func() C.int {
		return C.kyb_bn254_g2_msm(C.size_t(n), ptr(scalars), ptr(points), ptr(out), ptr(status), C.uint32_t(flags))
	})
	return out, status[:n], err
}

// Bn254G1Unmarshal / G2Unmarshal: batch UnmarshalBinary (coordinates < p, on the curve, G2 in the subgroup).
func Bn254G1Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("points", points, 64)
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g1_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)) })
	return
}

func Bn254G2Unmarshal(points []byte, flags uint32) (out, status []byte, err error) {
	n, err := count("points", points, 128)
	if err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g2_unmarshal(C.size_t(n), ptr(points), ptr(out), ptr(status), C.uint32_t(flags)) })
	return
}

func Bn254G1Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 64)
	if err = firstErr(err, need("b", b, n, 64)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g1_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bn254G2Add(a, b []byte) (out, status []byte, err error) {
	n, err := count("a", a, 128)
	if err = firstErr(err, need("b", b, n, 128)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 128*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_g2_add(C.size_t(n), ptr(a), ptr(b), ptr(out), ptr(status)) })
	return
}

func Bn254Pair(g1, g2 []byte, flags uint32) (gt, status []byte, err error) {
	n, err := count("g1", g1, 64)
	if err = firstErr(err, need("g2", g2, n, 128)); err != nil {
		return nil, nil, err
	}
	gt, status = make([]byte, 384*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_pair(C.size_t(n), ptr(g1), ptr(g2), ptr(gt), ptr(status), C.uint32_t(flags)) })
	return
}

func Bn254ValidatePairing(p1, p2, inv1, inv2 []byte, flags uint32) (ok, status []byte, err error) {
	n, err := count("p1", p1, 64)
	if err = firstErr(err, need("p2", p2, n, 128), need("inv1", inv1, n, 64), need("inv2", inv2, n, 128)); err != nil {
		return nil, nil, err
	}
	ok, status = make([]byte, n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn254_pair_check(C.size_t(n), ptr(p1), ptr(p2), ptr(inv1), ptr(inv2), ptr(ok), ptr(status), C.uint32_t(flags))
	})
	return
}

// Bn254GTMul: out[i] = gt[i]^scalars[i]  (pointGT.Mul, pairing/bn254/point.go:606 -> gfP12.Exp); coefficients >= p are rejected.
func Bn254GTMul(scalars, gt []byte) (out, status []byte, err error) {
	n, err := count("scalars", scalars, 32)
	if err = firstErr(err, need("gt", gt, n, 384)); err != nil {
		return nil, nil, err
	}
	out, status = make([]byte, 384*n), make([]byte, n)
	err = call(func() C.int { return C.kyb_bn254_gt_mul(C.size_t(n), ptr(scalars), ptr(gt), ptr(out), ptr(status)) })
	return
}

// Bn254HashG1: pointG1.Hash (pairing/bn254/point.go:207-285) for n messages of msgLen bytes under the suite's domain
// separation tag (suite.go:42-44, SetDomainG1).
func Bn254HashG1(n int, msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, nil, err
	}
	if len(dst) > 255 {
		return nil, nil, fmt.Errorf("kyberhip: domain separation tag of %d bytes (at most 255)", len(dst))
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn254_hash_g1(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out), ptr(status))
	})
	return
}

// Bn256HashG1: the package-level bn256.HashG1 (pairing/bn256/hash.go:10-110: HKDF-SHA-256 to the base field, then the
// Shallue-van de Woestijne map) for n messages of msgLen bytes; dst may be nil (hash_test.go:11-20).
func Bn256HashG1(n int, msgs []byte, msgLen int, dst []byte) (out, status []byte, err error) {
	if err = messages(msgs, msgLen, n); err != nil {
		return nil, nil, err
	}
	if len(dst) > 255 {
		return nil, nil, fmt.Errorf("kyberhip: domain separation tag of %d bytes (at most 255)", len(dst))
	}
	out, status = make([]byte, 64*n), make([]byte, n)
	err = call(func() C.int {
		return C.kyb_bn256_hash_g1_svdw(C.size_t(n), ptr(msgs), C.size_t(msgLen), ptr(dst), C.size_t(len(dst)), ptr(out), ptr(status))
	})
	return
}
