//go:build hip

package suite

import (
	"crypto/cipher"
	"hash"
	"io"
	"reflect"

	"go.dedis.ch/kyber/v4"
	"go.dedis.ch/kyber/v4/group/edwards25519"
	hip "go.dedis.ch/kyber/v4/hip"
	"go.dedis.ch/kyber/v4/pairing/bls12381/kilic"
	"go.dedis.ch/kyber/v4/pairing/bn254"
	"go.dedis.ch/kyber/v4/pairing/bn256"
)

// SuiteEd25519 is group/edwards25519's suite (suite.go:60) with the engine behind Point.Mul: a kyber.Group and a
// suites.Suite (Encoding, HashFactory, XOFFactory, Random delegated).
type SuiteEd25519 struct {
	*Group
	ref *edwards25519.SuiteEd25519
}

// NewBlakeSHA256Ed25519HIP mirrors edwards25519.NewBlakeSHA256Ed25519.
func NewBlakeSHA256Ed25519HIP() *SuiteEd25519 {
	ref := edwards25519.NewBlakeSHA256Ed25519()
	return &SuiteEd25519{Group: &Group{name: "Ed25519.hip", inner: ref, kind: hip.Ed25519}, ref: ref}
}

func (s *SuiteEd25519) String() string                        { return "Ed25519.hip" }
func (s *SuiteEd25519) Hash() hash.Hash                       { return s.ref.Hash() }
func (s *SuiteEd25519) XOF(seed []byte) kyber.XOF             { return s.ref.XOF(seed) }
func (s *SuiteEd25519) RandomStream() cipher.Stream           { return s.ref.RandomStream() }
func (s *SuiteEd25519) Read(r io.Reader, objs ...any) error   { return s.ref.Read(r, objs...) }
func (s *SuiteEd25519) Write(w io.Writer, objs ...any) error  { return s.ref.Write(w, objs...) }
func (s *SuiteEd25519) New(t reflect.Type) any                { return s.ref.New(t) }

// NewSuiteBLS12381 is the pairing suite over the reference's kilic adapter (kilic/suite.go:27): same encodings, same
// hash-to-curve domains, Pair / ValidatePairing / Mul on the device.
func NewSuiteBLS12381() *PairingSuite {
	ref := kilic.NewBLS12381Suite()
	return &PairingSuite{inner: ref, curve: curveBLS12381,
		g1: &Group{name: "bls12-381.G1.hip", inner: ref.G1(), kind: hip.Bls12381G1},
		g2: &Group{name: "bls12-381.G2.hip", inner: ref.G2(), kind: hip.Bls12381G2},
		gt: &Group{name: "bls12-381.GT.hip", inner: ref.GT(), kind: -1}}
}

// NewSuiteBn256 is the pairing suite over pairing/bn256 (suite.go:43).
func NewSuiteBn256() *PairingSuite {
	ref := bn256.NewSuite()
	return &PairingSuite{inner: ref, curve: curveBn256,
		g1: &Group{name: "bn256.G1.hip", inner: ref.G1(), kind: hip.Bn256G1},
		g2: &Group{name: "bn256.G2.hip", inner: ref.G2(), kind: hip.Bn256G2},
		gt: &Group{name: "bn256.GT.hip", inner: ref.GT(), kind: -1}}
}

// NewSuiteBn254 is the pairing suite over pairing/bn254 (Ethereum's alt_bn128; suite.go:52): the reference's points and
// scalars, its Keccak / Shallue-van de Woestijne Hash, Pair / ValidatePairing / Mul on the device.
func NewSuiteBn254() *PairingSuite {
	ref := bn254.NewSuite()
	return &PairingSuite{inner: ref, curve: curveBn254,
		g1: &Group{name: "bn254.G1.hip", inner: ref.G1(), kind: hip.Bn254G1},
		g2: &Group{name: "bn254.G2.hip", inner: ref.G2(), kind: hip.Bn254G2},
		gt: &Group{name: "bn254.GT.hip", inner: ref.GT(), kind: -1}}
}

// GroupSuite adapts a pairing suite to the single-group suites.Suite the registry and the benchmark take, the way
// kilic/adapter.go:18-51 does: points are public keys (G2), scalars and signatures live with G1.
type GroupSuite struct {
	*PairingSuite
	name string
}

var _ kyber.Group = (*GroupSuite)(nil)

func NewGroupSuiteBLS12381() *GroupSuite { return &GroupSuite{NewSuiteBLS12381(), "bls12-381.hip.adapter"} }
func NewGroupSuiteBn256() *GroupSuite    { return &GroupSuite{NewSuiteBn256(), "bn256.hip.adapter"} }
func NewGroupSuiteBn254() *GroupSuite    { return &GroupSuite{NewSuiteBn254(), "bn254.hip.adapter"} }

func (s *GroupSuite) Point() kyber.Point   { return s.G2().Point() }
func (s *GroupSuite) PointLen() int        { return s.G2().PointLen() }
func (s *GroupSuite) Scalar() kyber.Scalar { return s.G1().Scalar() }
func (s *GroupSuite) ScalarLen() int       { return s.G1().ScalarLen() }
func (s *GroupSuite) String() string       { return s.name }

// the batch face of the adapter is that of its point group (G2: public keys)
func (s *GroupSuite) BatchMul(scalars []kyber.Scalar, points []kyber.Point) ([]kyber.Point, error) {
	return s.g2.BatchMul(scalars, points)
}
func (s *GroupSuite) Commit(coeffs []kyber.Scalar, base kyber.Point) ([]kyber.Point, error) {
	return s.g2.Commit(coeffs, base)
}
func (s *GroupSuite) MSM(scalars []kyber.Scalar, points []kyber.Point, bits uint) (kyber.Point, error) {
	return s.g2.MSM(scalars, points, bits)
}
func (s *GroupSuite) Validate(encodings [][]byte) ([]byte, error) { return s.g2.Validate(encodings) }

var _ BatchGroup = (*GroupSuite)(nil)
