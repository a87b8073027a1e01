// Package suite is the drop-in form of the MI355X engine: kyber.Group / kyber.Point / pairing.Suite implementations
// for Ed25519, BLS12-381 and bn256 whose hot path -- Point.Mul, Suite.Pair, Suite.ValidatePairing and the batch
// methods found by type assertion (BatchGroup, BatchPairing) -- runs on the GPU through package kyberhip, and whose
// every other method is the reference suite's own (a Point here wraps the reference's point: group law, hashing,
// embedding, marshalling are delegated; scalars ARE the reference's scalars, host plumbing that never reaches the
// device except as 32 marshalled bytes).
//
// NOT COMPILED in the repository that ships it (no Go toolchain there): tests/test_cabi.py checks the method sets
// below against group.go:23-131, 175-183, encoding.go:15-32 and pairing/pairing.go:8-20 by regular expression.
//
// Variable-time: the engine's table loads are indexed by scalar digits and its inversions loop on their operand.
// These suites belong next to the other variable-time ones (suites/all_vartime.go), never under
// suites.RequireConstantTime.
//
//go:build hip

package suite

import (
	"crypto/cipher"
	"errors"
	"fmt"
	"io"

	"go.dedis.ch/kyber/v4"
	hip "go.dedis.ch/kyber/v4/hip"
)

// Point is a kyber.Point of one of the engine's groups.
type Point struct {
	g *Group
	p kyber.Point // the reference suite's point
}

var (
	_ kyber.Point         = (*Point)(nil)
	_ kyber.AllowsVarTime = (*Point)(nil)
)

// un returns the reference point behind x (x itself when it already is one).
func un(x kyber.Point) kyber.Point {
	if w, ok := x.(*Point); ok {
		return w.p
	}
	return x
}

func (P *Point) wrap(p kyber.Point) *Point { return &Point{g: P.g, p: p} }

// Inner exposes the reference point (for code that type-asserts on the reference's concrete types).
func (P *Point) Inner() kyber.Point { return P.p }

func (P *Point) Equal(s2 kyber.Point) bool { return P.p.Equal(un(s2)) }

func (P *Point) Null() kyber.Point {
	P.p = P.p.Null()
	return P
}

func (P *Point) Base() kyber.Point {
	P.p = P.p.Base()
	return P
}

func (P *Point) Pick(rand cipher.Stream) kyber.Point {
	P.p = P.p.Pick(rand)
	return P
}

func (P *Point) Set(p kyber.Point) kyber.Point {
	P.p = P.p.Set(un(p))
	return P
}

func (P *Point) Clone() kyber.Point { return P.wrap(P.p.Clone()) }

func (P *Point) EmbedLen() int { return P.p.EmbedLen() }

func (P *Point) Embed(data []byte, r cipher.Stream) kyber.Point {
	P.p = P.p.Embed(data, r)
	return P
}

func (P *Point) Data() ([]byte, error) { return P.p.Data() }

func (P *Point) Add(a, b kyber.Point) kyber.Point {
	P.p = P.p.Add(un(a), un(b))
	return P
}

func (P *Point) Sub(a, b kyber.Point) kyber.Point {
	P.p = P.p.Sub(un(a), un(b))
	return P
}

func (P *Point) Neg(a kyber.Point) kyber.Point {
	P.p = P.p.Neg(un(a))
	return P
}

// SingleOpOnDevice sends the single-element methods of the kyber interfaces -- Point.Mul, Suite.Pair,
// Suite.ValidatePairing -- to the device like a batch of one.  Off by default: ONE multiplication or pairing is a
// lone lane in a lone wave on a 256-CU chip, and the host-buffer round trip alone (upload, launch, download,
// synchronise: ~40-60 us; the arithmetic in a lone wave another 1-20 ms, profiles/r03_single_call_latency.json)
// exceeds what the reference needs on one core (0.06-0.35 ms per Mul, 1.6 ms per Pair; measured: 1.0-9.0 ms per Mul, 3.3-6.2 ms per Pair on the device).  SURVEY.md section 8b: keep
// n = 1 on the CPU.  The reference point embedded in every Point IS that CPU path, so delegating costs one branch and
// the result is the reference's own bytes by definition.  Tests of the device path through the single-element
// interface (util/test.CompareGroups) switch it on.
var SingleOpOnDevice = false

// MinDeviceBatch / MinDevicePairings are the batch sizes from which the batch methods (BatchMul, Commit, MSM; BatchPair,
// BatchValidatePairing) use the device; smaller batches loop over the reference on the CPU.  A device call costs
// what ONE wave costs whatever the batch (up to 64 lanes x 4 SIMDs x 256 CUs elements run side by side) -- measured on
// MI355X through the host-buffer entry points (profiles/r03_single_call_latency.json, median of 9 calls): 1.0 ms for an
// Ed25519 Mul, 3.7 / 9.0 ms for a BLS12-381 G1 / G2 Mul, 1.6 ms for a bn256 G1 Mul, 6.2 / 3.3 ms for a BLS12-381 /
// bn256 Pair, flat from 1 to 4096 elements -- so the break-even against one CPU core of the reference (0.35 ms,
// ~0.12 / 0.28 ms, 0.15 ms, 1.6 ms) is 4 elements for Ed25519 and the pairings and 16-64 for the pairing-curve
// multiplications.
var (
	MinDeviceBatch    = 64
	MinDevicePairings = 8
)

// Mul: P = s * p (p == nil: the standard base).  One multiplication per call is the reference's signature and a
// single element stays on the CPU (SingleOpOnDevice); loops belong in Group.BatchMul / Commit / MSM, which is where the
// device is.  On the device the engine reproduces the reference's result bytes, including group/edwards25519's
// treatment of scalars >= 2^255 (SURVEY.md 8a); an engine failure panics, as the reference's type-cast failures do.
func (P *Point) Mul(s kyber.Scalar, p kyber.Point) kyber.Point {
	if !SingleOpOnDevice {
		if p == nil {
			P.p = P.p.Mul(s, nil)
		} else {
			P.p = P.p.Mul(s, un(p))
		}
		return P
	}
	return P.mulOnDevice(s, p)
}

func (P *Point) mulOnDevice(s kyber.Scalar, p kyber.Point) kyber.Point {
	sb, err := s.MarshalBinary()
	if err != nil {
		panic(err)
	}
	var out, st []byte
	if p == nil {
		out, st, err = P.g.mulBase(sb)
	} else {
		var pb []byte
		if pb, err = un(p).MarshalBinary(); err != nil {
			panic(err)
		}
		out, st, err = P.g.mul(sb, pb)
	}
	if err == nil && len(st) == 1 && st[0] != 0 {
		err = fmt.Errorf("kyberhip: %s: operand rejected (status %d)", P.g.name, st[0])
	}
	if err == nil {
		err = P.p.UnmarshalBinary(out)
	}
	if err != nil {
		panic(err)
	}
	return P
}

// AllowVarTime: the engine is variable-time in any case; the flag only selects, for Ed25519, the value semantics of
// geScalarMultVartime (every scalar bit counts) over geScalarMult's.
func (P *Point) AllowVarTime(on bool) {
	P.g.vartime = on
	if a, ok := P.p.(kyber.AllowsVarTime); ok {
		a.AllowVarTime(on)
	}
}

// ---- kyber.Marshaling (encoding.go:15-32): the reference's own encodings
func (P *Point) MarshalBinary() ([]byte, error) { return P.p.MarshalBinary() }
func (P *Point) UnmarshalBinary(b []byte) error { return P.p.UnmarshalBinary(b) }
func (P *Point) String() string                 { return P.p.String() }
func (P *Point) MarshalSize() int               { return P.p.MarshalSize() }
func (P *Point) MarshalTo(w io.Writer) (int, error) {
	return P.p.MarshalTo(w)
}
func (P *Point) UnmarshalFrom(r io.Reader) (int, error) {
	return P.p.UnmarshalFrom(r)
}

// Hash implements kyber.HashablePoint where the reference point does (BLS12-381 G1 / G2, bn256 G1, Ed25519).
func (P *Point) Hash(m []byte) kyber.Point {
	h, ok := P.p.(kyber.HashablePoint)
	if !ok {
		panic("kyberhip: " + P.g.name + ": point is not hashable")
	}
	P.p = h.Hash(m)
	return P
}

// IsInCorrectGroup implements kyber.SubGroupElement where the reference point does.
func (P *Point) IsInCorrectGroup() bool {
	if e, ok := P.p.(kyber.SubGroupElement); ok {
		return e.IsInCorrectGroup()
	}
	return true
}

var errLen = errors.New("kyberhip: slices of different length")

// encodings marshals points (unwrapped) back to back.
func encodings(points []kyber.Point, size int) ([]byte, error) {
	buf := make([]byte, 0, size*len(points))
	for _, x := range points {
		b, err := un(x).MarshalBinary()
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

func scalarBytes(scalars []kyber.Scalar) ([]byte, error) {
	buf := make([]byte, 0, 32*len(scalars))
	for _, s := range scalars {
		b, err := s.MarshalBinary()
		if err != nil {
			return nil, err
		}
		if len(b) != 32 {
			return nil, fmt.Errorf("kyberhip: %d-byte scalar, want 32", len(b))
		}
		buf = append(buf, b...)
	}
	return buf, nil
}

func firstBad(status []byte) error {
	for i, s := range status {
		if s != 0 {
			return fmt.Errorf("kyberhip: element %d does not unmarshal (status %d)", i, s)
		}
	}
	return nil
}

var _ = hip.Vartime
