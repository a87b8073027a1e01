//go:build hip

package suite

import (
	"crypto/cipher"
	"fmt"
	"hash"
	"io"
	"reflect"

	"go.dedis.ch/kyber/v4"
	hip "go.dedis.ch/kyber/v4/hip"
	"go.dedis.ch/kyber/v4/pairing"
)

// PairingSuite is a pairing.Suite (pairing/pairing.go:8-20) whose Pair / ValidatePairing and G1 / G2 multiplications
// run on the device.  GT elements are the reference's (GT arithmetic is not on the hot path; GTElt.Mul has a batch
// form, BatchGTMul).
type PairingSuite struct {
	inner      pairing.Suite
	g1, g2, gt *Group
	curve      curveID // which engine entry points: BLS12-381 (flags, fused verification), bn256, bn254 (flags)
}

type curveID int

const (
	curveBLS12381 curveID = iota
	curveBn256
	curveBn254
)

var _ pairing.Suite = (*PairingSuite)(nil)

func (s *PairingSuite) G1() kyber.Group { return s.g1 }
func (s *PairingSuite) G2() kyber.Group { return s.g2 }
func (s *PairingSuite) GT() kyber.Group { return s.gt }

// Pair computes e(p1, p2) and returns it as a GT point of this suite: ONE pairing is the reference's own (the embedded
// suite, one CPU core, 1.6 ms -- a lone wave of the tower machine needs ~20 ms for it), unless SingleOpOnDevice asks
// for the device path; batches go through BatchPair.
func (s *PairingSuite) Pair(p1, p2 kyber.Point) kyber.Point {
	if !SingleOpOnDevice {
		return &Point{g: s.gt, p: s.inner.Pair(un(p1), un(p2))}
	}
	out, err := s.pairOnDevice([]kyber.Point{p1}, []kyber.Point{p2})
	if err != nil {
		panic(err)
	}
	return out[0]
}

// ValidatePairing reports e(p1, p2) == e(inv1, inv2); a single check is the reference's (see Pair).
func (s *PairingSuite) ValidatePairing(p1, p2, inv1, inv2 kyber.Point) bool {
	if !SingleOpOnDevice {
		return s.inner.ValidatePairing(un(p1), un(p2), un(inv1), un(inv2))
	}
	ok, err := s.validateOnDevice([]kyber.Point{p1}, []kyber.Point{p2}, []kyber.Point{inv1}, []kyber.Point{inv2})
	if err != nil {
		panic(err)
	}
	return ok[0]
}

// ---- kyber.Encoding, HashFactory, XOFFactory, Random: the reference suite's
func (s *PairingSuite) New(t reflect.Type) any                { return s.inner.New(t) }
func (s *PairingSuite) Read(r io.Reader, objs ...any) error   { return s.inner.Read(r, objs...) }
func (s *PairingSuite) Write(w io.Writer, objs ...any) error  { return s.inner.Write(w, objs...) }
func (s *PairingSuite) Hash() hash.Hash                       { return s.inner.Hash() }
func (s *PairingSuite) XOF(seed []byte) kyber.XOF             { return s.inner.XOF(seed) }
func (s *PairingSuite) RandomStream() cipher.Stream           { return s.inner.RandomStream() }

// BatchPairing is what callers type-assert a pairing.Suite to.
type BatchPairing interface {
	pairing.Suite
	BatchPair(p1, p2 []kyber.Point) ([]kyber.Point, error)
	BatchValidatePairing(p1, p2, inv1, inv2 []kyber.Point) ([]bool, error)
}

var _ BatchPairing = (*PairingSuite)(nil)

func (s *PairingSuite) BatchPair(p1, p2 []kyber.Point) ([]kyber.Point, error) {
	if len(p1) != len(p2) {
		return nil, errLen
	}
	if len(p1) < MinDevicePairings && !SingleOpOnDevice { // too few for one wave to beat a CPU core: the reference, in a loop
		out := make([]kyber.Point, len(p1))
		for i := range out {
			out[i] = &Point{g: s.gt, p: s.inner.Pair(un(p1[i]), un(p2[i]))}
		}
		return out, nil
	}
	return s.pairOnDevice(p1, p2)
}

func (s *PairingSuite) pairOnDevice(p1, p2 []kyber.Point) ([]kyber.Point, error) {
	a, err := encodings(p1, s.g1.pointLen())
	if err != nil {
		return nil, err
	}
	b, err := encodings(p2, s.g2.pointLen())
	if err != nil {
		return nil, err
	}
	var gt, st []byte
	size := 384
	switch s.curve {
	case curveBLS12381:
		size = 576
		gt, st, err = hip.Bls12381Pair(a, b, hip.Trusted(0)|hip.Trusted(1)) // kyber.Points: validated when unmarshalled
	case curveBn254:
		gt, st, err = hip.Bn254Pair(a, b, hip.Trusted(0)|hip.Trusted(1))
	default:
		gt, st, err = hip.Bn256Pair(a, b)
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	out := make([]kyber.Point, len(p1))
	for i := range out {
		p := s.gt.inner.Point()
		if err = p.UnmarshalBinary(gt[i*size : (i+1)*size]); err != nil {
			return nil, err
		}
		out[i] = &Point{g: s.gt, p: p}
	}
	return out, nil
}

func (s *PairingSuite) BatchValidatePairing(p1, p2, inv1, inv2 []kyber.Point) ([]bool, error) {
	if len(p1) == len(p2) && len(p1) == len(inv1) && len(p1) == len(inv2) && len(p1) < MinDevicePairings && !SingleOpOnDevice {
		ok := make([]bool, len(p1))
		for i := range ok {
			ok[i] = s.inner.ValidatePairing(un(p1[i]), un(p2[i]), un(inv1[i]), un(inv2[i]))
		}
		return ok, nil
	}
	return s.validateOnDevice(p1, p2, inv1, inv2)
}

func (s *PairingSuite) validateOnDevice(p1, p2, inv1, inv2 []kyber.Point) ([]bool, error) {
	n := len(p1)
	if len(p2) != n || len(inv1) != n || len(inv2) != n {
		return nil, errLen
	}
	a, err := encodings(p1, s.g1.pointLen())
	if err != nil {
		return nil, err
	}
	b, err := encodings(p2, s.g2.pointLen())
	if err != nil {
		return nil, err
	}
	c, err := encodings(inv1, s.g1.pointLen())
	if err != nil {
		return nil, err
	}
	d, err := encodings(inv2, s.g2.pointLen())
	if err != nil {
		return nil, err
	}
	var ok, st []byte
	switch s.curve {
	case curveBLS12381:
		ok, st, err = hip.Bls12381ValidatePairing(a, b, c, d, hip.TrustedAll)
	case curveBn254:
		ok, st, err = hip.Bn254ValidatePairing(a, b, c, d, hip.TrustedAll)
	default:
		ok, st, err = hip.Bn256ValidatePairing(a, b, c, d)
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	res := make([]bool, n)
	for i := range res {
		res[i] = ok[i] == 1
	}
	return res, nil
}

// BatchVerify is N x sign/bls Verify (bls.go:82-96) for the scheme with signatures on G1 and keys on G2, messages of
// equal length: hash-to-curve, both unmarshal checks, the product of two Miller loops and one final exponentiation
// per signature, all on the device.  Public keys are kyber.Points (validated); signatures are the raw bytes received.
// A signature of the wrong length verifies false by itself.  BLS12-381 only.
func (s *PairingSuite) BatchVerify(publics []kyber.Point, msgs [][]byte, sigs [][]byte, dst []byte) ([]bool, error) {
	if s.curve != curveBLS12381 {
		return nil, fmt.Errorf("kyberhip: fused verification exists for BLS12-381 only")
	}
	n := len(msgs)
	if len(publics) != n || len(sigs) != n {
		return nil, errLen
	}
	res := make([]bool, n)
	if n == 0 {
		return res, nil
	}
	pk, err := encodings(publics, 96)
	if err != nil {
		return nil, err
	}
	ml := len(msgs[0])
	mb := make([]byte, 0, n*ml)
	sb := make([]byte, 0, n*48)
	short := make([]bool, n)
	for i := range msgs {
		if len(msgs[i]) != ml {
			return nil, fmt.Errorf("kyberhip: messages of different length (group them by length)")
		}
		mb = append(mb, msgs[i]...)
		sg := sigs[i]
		if len(sg) != 48 {
			short[i] = true
			sg = make([]byte, 48)
		}
		sb = append(sb, sg...)
	}
	ok, st, err := hip.Bls12381VerifyG1(pk, mb, ml, dst, sb, hip.Trusted(0))
	if err != nil {
		return nil, err
	}
	for i := range res {
		res[i] = ok[i] == 1 && st[i] == 0 && !short[i]
	}
	return res, nil
}

// BatchVerifySameMsg is the verification loop of tbls.Recover (sign/tbls/tbls.go:118-131): ONE message, a different
// public key per signature (the public shares public.Eval(idx).V).  H(msg) is hashed once per call on the device
// (kyb_bls12381_verify_g1_same_msg); otherwise BatchVerify.  BLS12-381 only.
func (s *PairingSuite) BatchVerifySameMsg(publics []kyber.Point, msg []byte, sigs [][]byte, dst []byte) ([]bool, error) {
	if s.curve != curveBLS12381 {
		return nil, fmt.Errorf("kyberhip: fused verification exists for BLS12-381 only")
	}
	n := len(sigs)
	if len(publics) != n {
		return nil, errLen
	}
	res := make([]bool, n)
	if n == 0 {
		return res, nil
	}
	pk, err := encodings(publics, 96)
	if err != nil {
		return nil, err
	}
	sb := make([]byte, 0, n*48)
	short := make([]bool, n)
	for i, sg := range sigs {
		if len(sg) != 48 {
			short[i] = true
			sg = make([]byte, 48)
		}
		sb = append(sb, sg...)
	}
	ok, st, err := hip.Bls12381VerifyG1SameMsg(pk, msg, dst, sb, hip.Trusted(0))
	if err != nil {
		return nil, err
	}
	for i := range res {
		res[i] = ok[i] == 1 && st[i] == 0 && !short[i]
	}
	return res, nil
}

// BatchVerifySameKey is sign/bls Verify (bls.go:82-96) in a loop with ONE public key -- a drand chain, one tbls
// participant's partial signatures: both Miller loops from line tables (kyb_bls12381_verify_g1_same_key; the tables of
// the last eight keys are kept per stream, so a committee of a few keys pays the table walk once per key).
func (s *PairingSuite) BatchVerifySameKey(public kyber.Point, msgs [][]byte, sigs [][]byte, dst []byte) ([]bool, error) {
	if s.curve != curveBLS12381 {
		return nil, fmt.Errorf("kyberhip: fused verification exists for BLS12-381 only")
	}
	n := len(msgs)
	if len(sigs) != n {
		return nil, errLen
	}
	res := make([]bool, n)
	if n == 0 {
		return res, nil
	}
	pk, err := encodings([]kyber.Point{public}, 96)
	if err != nil {
		return nil, err
	}
	ml := len(msgs[0])
	mb := make([]byte, 0, n*ml)
	sb := make([]byte, 0, n*48)
	short := make([]bool, n)
	for i := range msgs {
		if len(msgs[i]) != ml {
			return nil, fmt.Errorf("kyberhip: messages of different length (group them by length)")
		}
		mb = append(mb, msgs[i]...)
		sg := sigs[i]
		if len(sg) != 48 {
			short[i] = true
			sg = make([]byte, 48)
		}
		sb = append(sb, sg...)
	}
	ok, st, err := hip.Bls12381VerifyG1SameKey(pk, mb, ml, dst, sb, hip.Trusted(0))
	if err != nil {
		return nil, err
	}
	for i := range res {
		res[i] = ok[i] == 1 && st[i] == 0 && !short[i]
	}
	return res, nil
}

// BatchGTMul: out[i] = gt[i]^scalars[i].
func (s *PairingSuite) BatchGTMul(scalars []kyber.Scalar, gts []kyber.Point) ([]kyber.Point, error) {
	if len(scalars) != len(gts) {
		return nil, errLen
	}
	size := 384
	if s.curve == curveBLS12381 {
		size = 576
	}
	sb, err := scalarBytes(scalars)
	if err != nil {
		return nil, err
	}
	gb, err := encodings(gts, size)
	if err != nil {
		return nil, err
	}
	var out, st []byte
	switch s.curve {
	case curveBLS12381:
		out, st, err = hip.Bls12381GTMul(sb, gb)
	case curveBn254:
		out, st, err = hip.Bn254GTMul(sb, gb)
	default:
		out, st, err = hip.Bn256GTMul(sb, gb)
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	res := make([]kyber.Point, len(gts))
	for i := range res {
		p := s.gt.inner.Point()
		if err = p.UnmarshalBinary(out[i*size : (i+1)*size]); err != nil {
			return nil, err
		}
		res[i] = &Point{g: s.gt, p: p}
	}
	return res, nil
}
