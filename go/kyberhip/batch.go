// Batch helpers over kyber's own interfaces: they marshal []kyber.Scalar / []kyber.Point of ANY suite whose
// encodings are the reference's (group/edwards25519, pairing/bls12381/*, pairing/bn256, pairing/bn254), run one engine call, and
// unmarshal the results back into points of that same suite -- so every other method of the returned values is the
// reference's own.  This is the smallest possible integration: the MSM-shaped loops of share/poly.go and
// sign/bdn/bdn.go call these instead of n x (Mul + Add); see INTEGRATION.md for the patches.
//
// NOT COMPILED in the repository that ships it (no Go toolchain there).
//
//go:build hip

package kyberhip

import (
	"errors"
	"fmt"

	"go.dedis.ch/kyber/v4"
)

// Kind selects the engine entry points for a kyber.Group.
type Kind int

const (
	Ed25519 Kind = iota
	Bls12381G1
	Bls12381G2
	Bn256G1
	Bn256G2
	Bn254G1
	Bn254G2
)

func marshalAll[T interface{ MarshalBinary() ([]byte, error) }](xs []T, size int) ([]byte, error) {
	buf := make([]byte, 0, size*len(xs))
	for _, x := range xs {
		b, err := x.MarshalBinary()
		if err != nil {
			return nil, err
		}
		if len(b) != size {
			return nil, fmt.Errorf("kyberhip: %d-byte encoding, want %d", len(b), size)
		}
		buf = append(buf, b...)
	}
	return buf, nil
}

func pointLen(k Kind) int { return [...]int{32, 48, 96, 64, 128}[k] }

func firstBad(status []byte) error {
	for i, s := range status {
		if s != 0 {
			return fmt.Errorf("kyberhip: element %d does not unmarshal (status %d)", i, s)
		}
	}
	return nil
}

// points handed in as kyber.Point values were validated when they were unmarshalled: the engine need not re-check
func trusted(k Kind) uint32 {
	if k == Bls12381G1 || k == Bls12381G2 {
		return Trusted(0)
	}
	return 0
}

// BatchMul returns out[i] = scalars[i] * points[i] as points of group g (N x Point.Mul, group.go:128-130).
func BatchMul(k Kind, g kyber.Group, scalars []kyber.Scalar, points []kyber.Point) ([]kyber.Point, error) {
	if len(scalars) != len(points) {
		return nil, errors.New("kyberhip: length mismatch")
	}
	sb, err := marshalAll(scalars, 32)
	if err != nil {
		return nil, err
	}
	pb, err := marshalAll(points, pointLen(k))
	if err != nil {
		return nil, err
	}
	var out, st []byte
	switch k {
	case Ed25519:
		out, st, err = Ed25519Mul(sb, pb, 0)
	case Bls12381G1:
		out, st, err = Bls12381G1Mul(sb, pb, trusted(k))
	case Bls12381G2:
		out, st, err = Bls12381G2Mul(sb, pb, trusted(k))
	case Bn256G1:
		out, st, err = Bn256G1Mul(sb, pb)
	case Bn256G2:
		out, st, err = Bn256G2Mul(sb, pb)
	case Bn254G1:
		out, st, err = Bn254G1Mul(sb, pb, Trusted(0)) // kyber.Points: unmarshalled (validated) before
	case Bn254G2:
		out, st, err = Bn254G2Mul(sb, pb, Trusted(0))
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	return unmarshalAll(g, out, pointLen(k))
}

// MSM returns sum_i scalars[i] * points[i]: what PubPoly.Eval (share/poly.go:340-348), RecoverCommit (:449-476) and
// bdn.AggregateSignatures / AggregatePublicKeys (sign/bdn/bdn.go:126-181) compute with n x (Mul + Add).
func MSM(k Kind, g kyber.Group, scalars []kyber.Scalar, points []kyber.Point) (kyber.Point, error) {
	return MSMBits(k, g, scalars, points, 0)
}

// MSMBits is MSM for scalars known to be below 2^bits (sign/bdn's coefficients: bits = 129); bits = 0: full length.
func MSMBits(k Kind, g kyber.Group, scalars []kyber.Scalar, points []kyber.Point, bits uint) (kyber.Point, error) {
	fl := uint32(0)
	if bits != 0 {
		fl = ScalarBits(bits)
	}
	if len(scalars) != len(points) {
		return nil, errors.New("kyberhip: length mismatch")
	}
	sb, err := marshalAll(scalars, 32)
	if err != nil {
		return nil, err
	}
	pb, err := marshalAll(points, pointLen(k))
	if err != nil {
		return nil, err
	}
	var out, st []byte
	switch k {
	case Ed25519:
		out, st, err = Ed25519MSM(sb, pb, fl)
	case Bls12381G1:
		out, st, err = Bls12381G1MSM(sb, pb, trusted(k)|fl)
	case Bls12381G2:
		out, st, err = Bls12381G2MSM(sb, pb, trusted(k)|fl)
	case Bn256G1:
		out, st, err = Bn256G1MSM(sb, pb, fl)
	case Bn256G2:
		out, st, err = Bn256G2MSM(sb, pb, fl)
	case Bn254G1:
		out, st, err = Bn254G1MSM(sb, pb, fl|Trusted(0))
	case Bn254G2:
		out, st, err = Bn254G2MSM(sb, pb, fl|Trusted(0))
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	p := g.Point()
	return p, p.UnmarshalBinary(out)
}

// Commit returns coeffs[i] * base: the loop of share.PriPoly.Commit (share/poly.go:143-149) in one call -- on the pairing
// suites through a fixed-base table of the base's multiples (26 table additions per coefficient instead of a ladder:
// built per call from 2^17 coefficients, kept on the device for the generator and for a repeated base from 64); base == nil is the
// group's standard base, as poly.go:144 passes nil through.
func Commit(k Kind, g kyber.Group, coeffs []kyber.Scalar, base kyber.Point) ([]kyber.Point, error) {
	sb, err := marshalAll(coeffs, 32)
	if err != nil {
		return nil, err
	}
	if base == nil {
		base = g.Point().Base()
	}
	bb, err := base.MarshalBinary()
	if err != nil {
		return nil, err
	}
	var out, st []byte
	switch k {
	case Ed25519:
		out, st, err = Ed25519MulSameBase(sb, bb, 0)
	case Bls12381G1:
		out, st, err = Bls12381G1MulSameBase(sb, bb, trusted(k))
	case Bls12381G2:
		out, st, err = Bls12381G2MulSameBase(sb, bb, trusted(k))
	case Bn256G1:
		out, st, err = Bn256G1MulSameBase(sb, bb)
	case Bn256G2:
		out, st, err = Bn256G2MulSameBase(sb, bb)
	case Bn254G1:
		out, st, err = Bn254G1MulSameBase(sb, bb, Trusted(0))
	case Bn254G2:
		out, st, err = Bn254G2MulSameBase(sb, bb, Trusted(0))
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	return unmarshalAll(g, out, pointLen(k))
}

func unmarshalAll(g kyber.Group, buf []byte, size int) ([]kyber.Point, error) {
	out := make([]kyber.Point, len(buf)/size)
	for i := range out {
		out[i] = g.Point()
		if err := out[i].UnmarshalBinary(buf[i*size : (i+1)*size]); err != nil {
			return nil, err
		}
	}
	return out, nil
}

// BatchValidate runs N x Point.UnmarshalBinary on the device and reports, per element, whether the reference would
// accept the encoding (0) or return an error (non-zero): ZCash flag rules, on-curve and subgroup checks for
// BLS12-381; on-curve for bn256; "has a square root" for Ed25519.  The engine-backed suite of INTEGRATION.md
// section 3 stores encodings, so a batch of received points (DKG deals, bdn public keys, beacon signatures) is
// validated by ONE call and carried as Trusted afterwards.
func BatchValidate(k Kind, encodings []byte) (status []byte, err error) {
	if len(encodings)%pointLen(k) != 0 {
		return nil, errors.New("kyberhip: ragged encoding buffer")
	}
	switch k {
	case Ed25519:
		_, status, err = Ed25519Unmarshal(encodings)
	case Bls12381G1:
		_, status, err = Bls12381G1Unmarshal(encodings, 0)
	case Bls12381G2:
		_, status, err = Bls12381G2Unmarshal(encodings, 0)
	case Bn256G1:
		_, status, err = Bn256G1Unmarshal(encodings)
	case Bn256G2:
		_, status, err = Bn256G2Unmarshal(encodings)
	case Bn254G1:
		_, status, err = Bn254G1Unmarshal(encodings, 0) // bytes from the wire: full validation
	case Bn254G2:
		_, status, err = Bn254G2Unmarshal(encodings, 0)
	}
	return status, err
}
