// The reference's own conformance and parity harness over the engine's suites (util/test/test.go:325-427): GroupTest on
// every group, CompareGroups against the CPU suite it wraps -- once with the single-element policy as shipped (Mul / Pair
// delegate to the embedded reference: the comparison is then trivially equal and guards the delegation itself) and
// once with SingleOpOnDevice, which sends every Point.Mul of testGroup through the C ABI to the GPU.
//
// NOT COMPILED in the repository that ships it (no Go toolchain there); needs `-tags hip` and libkyberhip.so.
//
//go:build hip

package suite

import (
	"testing"

	"go.dedis.ch/kyber/v4"
	"go.dedis.ch/kyber/v4/group/edwards25519"
	"go.dedis.ch/kyber/v4/pairing"
	bls "go.dedis.ch/kyber/v4/pairing/bls12381/kilic"
	"go.dedis.ch/kyber/v4/pairing/bn254"
	"go.dedis.ch/kyber/v4/pairing/bn256"
	"go.dedis.ch/kyber/v4/util/test"
	"go.dedis.ch/kyber/v4/xof/blake2xb"
)

func pairs() []struct {
	name     string
	hip, ref kyber.Group
} {
	return []struct {
		name     string
		hip, ref kyber.Group
	}{
		{"ed25519", NewBlakeSHA256Ed25519HIP(), edwards25519.NewBlakeSHA256Ed25519()},
		{"bls12381.G1", NewSuiteBLS12381().G1(), bls.NewBLS12381Suite().G1()},
		{"bls12381.G2", NewSuiteBLS12381().G2(), bls.NewBLS12381Suite().G2()},
		{"bn256.G1", NewSuiteBn256().G1(), bn256.NewSuite().G1()},
		{"bn256.G2", NewSuiteBn256().G2(), bn256.NewSuite().G2()},
		{"bn254.G1", NewSuiteBn254().G1(), bn254.NewSuite().G1()},
		{"bn254.G2", NewSuiteBn254().G2(), bn254.NewSuite().G2()},
	}
}

func TestGroupConformanceAndParity(t *testing.T) {
	for _, onDevice := range []bool{false, true} {
		SingleOpOnDevice = onDevice
		for _, p := range pairs() {
			t.Run(p.name, func(t *testing.T) {
				test.GroupTest(t, p.hip)
				test.CompareGroups(t, blake2xb.New, p.hip, p.ref)
			})
		}
	}
	SingleOpOnDevice = false
}

// Pair / ValidatePairing: bilinearity through the suite interface, shipped policy and device path (pairing suites of
// the reference: pairing/bn256/suite_test.go:231-259).
func TestPairingThroughTheInterface(t *testing.T) {
	for _, onDevice := range []bool{false, true} {
		SingleOpOnDevice = onDevice
		for _, s := range []pairing.Suite{NewSuiteBLS12381(), NewSuiteBn256(), NewSuiteBn254()} {
			a, b := s.G1().Scalar().Pick(s.RandomStream()), s.G1().Scalar().Pick(s.RandomStream())
			ab := s.G1().Scalar().Mul(a, b)
			lhs := s.Pair(s.G1().Point().Mul(a, nil), s.G2().Point().Mul(b, nil))
			rhs := s.Pair(s.G1().Point().Mul(ab, nil), s.G2().Point().Base())
			if !lhs.Equal(rhs) {
				t.Fatalf("%v: e(aP, bQ) != e(abP, Q) (onDevice=%v)", s, onDevice)
			}
			if !s.ValidatePairing(s.G1().Point().Mul(a, nil), s.G2().Point().Mul(b, nil), s.G1().Point().Mul(ab, nil), s.G2().Point().Base()) {
				t.Fatalf("%v: ValidatePairing rejected a valid quadruple (onDevice=%v)", s, onDevice)
			}
		}
	}
	SingleOpOnDevice = false
}
