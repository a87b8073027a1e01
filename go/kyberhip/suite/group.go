//go:build hip

package suite

import (
	"crypto/cipher"
	"fmt"
	"hash"

	"go.dedis.ch/kyber/v4"
	hip "go.dedis.ch/kyber/v4/hip"
)

// Group is a kyber.Group (group.go:175-183) whose points multiply on the device.
type Group struct {
	name    string
	inner   kyber.Group // the reference group: scalars, point construction, lengths
	kind    hip.Kind
	vartime bool
}

var _ kyber.Group = (*Group)(nil)

func (g *Group) String() string       { return g.name }
func (g *Group) ScalarLen() int       { return g.inner.ScalarLen() }
func (g *Group) Scalar() kyber.Scalar { return g.inner.Scalar() }
func (g *Group) PointLen() int        { return g.inner.PointLen() }
func (g *Group) Point() kyber.Point   { return &Point{g: g, p: g.inner.Point()} }

// the optional faces the reference groups show (kyber.HashFactory, XOFFactory, Random), passed through
func (g *Group) Hash() hash.Hash {
	return g.inner.(kyber.HashFactory).Hash()
}
func (g *Group) XOF(seed []byte) kyber.XOF {
	return g.inner.(kyber.XOFFactory).XOF(seed)
}
func (g *Group) RandomStream() cipher.Stream {
	return g.inner.(kyber.Random).RandomStream()
}

func (g *Group) flags() uint32 {
	if g.kind == hip.Ed25519 && g.vartime {
		return hip.Vartime
	}
	return 0
}

// points that exist as kyber.Point values were validated when they were unmarshalled: no re-check on the device
func (g *Group) trusted() uint32 {
	if g.kind == hip.Bls12381G1 || g.kind == hip.Bls12381G2 || g.kind == hip.Bn254G1 || g.kind == hip.Bn254G2 {
		return hip.Trusted(0)
	}
	return 0
}

func (g *Group) pointLen() int { return [...]int{32, 48, 96, 64, 128, 64, 128}[g.kind] }

// mul: n scalars x n points (encodings back to back)
func (g *Group) mul(sb, pb []byte) (out, st []byte, err error) {
	switch g.kind {
	case hip.Ed25519:
		return hip.Ed25519Mul(sb, pb, g.flags())
	case hip.Bls12381G1:
		return hip.Bls12381G1Mul(sb, pb, g.trusted())
	case hip.Bls12381G2:
		return hip.Bls12381G2Mul(sb, pb, g.trusted())
	case hip.Bn256G1:
		return hip.Bn256G1Mul(sb, pb)
	case hip.Bn256G2:
		return hip.Bn256G2Mul(sb, pb)
	case hip.Bn254G1:
		return hip.Bn254G1Mul(sb, pb, g.trusted())
	case hip.Bn254G2:
		return hip.Bn254G2Mul(sb, pb, g.trusted())
	}
	return nil, nil, fmt.Errorf("kyberhip: %s has no device multiplication", g.name)
}

// mulBase: n scalars x the standard base
func (g *Group) mulBase(sb []byte) (out, st []byte, err error) {
	if g.kind == hip.Ed25519 {
		out, err = hip.Ed25519MulBase(sb, g.flags())
		return out, make([]byte, len(sb)/32), err
	}
	bb, err := g.inner.Point().Base().MarshalBinary()
	if err != nil {
		return nil, nil, err
	}
	return g.mulSame(sb, bb)
}

func (g *Group) mulSame(sb, bb []byte) (out, st []byte, err error) {
	switch g.kind {
	case hip.Ed25519:
		return hip.Ed25519MulSameBase(sb, bb, g.flags())
	case hip.Bls12381G1:
		return hip.Bls12381G1MulSameBase(sb, bb, g.trusted())
	case hip.Bls12381G2:
		return hip.Bls12381G2MulSameBase(sb, bb, g.trusted())
	case hip.Bn256G1:
		return hip.Bn256G1MulSameBase(sb, bb)
	case hip.Bn256G2:
		return hip.Bn256G2MulSameBase(sb, bb)
	case hip.Bn254G1:
		return hip.Bn254G1MulSameBase(sb, bb, g.trusted())
	case hip.Bn254G2:
		return hip.Bn254G2MulSameBase(sb, bb, g.trusted())
	}
	return nil, nil, fmt.Errorf("kyberhip: %s has no device multiplication", g.name)
}

func (g *Group) decode(buf []byte) ([]kyber.Point, error) {
	size := g.pointLen()
	out := make([]kyber.Point, len(buf)/size)
	for i := range out {
		p := g.inner.Point()
		if err := p.UnmarshalBinary(buf[i*size : (i+1)*size]); err != nil {
			return nil, err
		}
		out[i] = &Point{g: g, p: p}
	}
	return out, nil
}

// BatchGroup is what callers type-assert a kyber.Group to in order to replace their Mul loops.
type BatchGroup interface {
	kyber.Group
	// BatchMul: out[i] = scalars[i] * points[i]  (N x Point.Mul, group.go:128-130)
	BatchMul(scalars []kyber.Scalar, points []kyber.Point) ([]kyber.Point, error)
	// Commit: out[i] = coeffs[i] * base, base == nil: the standard base  (share.PriPoly.Commit, share/poly.go:143-149)
	Commit(coeffs []kyber.Scalar, base kyber.Point) ([]kyber.Point, error)
	// MSM: sum_i scalars[i] * points[i]  (PubPoly.Eval share/poly.go:340-348, RecoverCommit :449-476,
	// bdn.AggregateSignatures / AggregatePublicKeys sign/bdn/bdn.go:126-181); bits != 0: every scalar < 2^bits
	MSM(scalars []kyber.Scalar, points []kyber.Point, bits uint) (kyber.Point, error)
	// Validate: status[i] != 0 where Point.UnmarshalBinary(encodings[i]) returns an error
	Validate(encodings [][]byte) ([]byte, error)
}

var _ BatchGroup = (*Group)(nil)

func (g *Group) BatchMul(scalars []kyber.Scalar, points []kyber.Point) ([]kyber.Point, error) {
	if len(scalars) != len(points) {
		return nil, errLen
	}
	if len(scalars) < MinDeviceBatch && !SingleOpOnDevice { // below one wave's break-even: the reference, in a loop
		out := make([]kyber.Point, len(scalars))
		for i := range out {
			out[i] = &Point{g: g, p: g.inner.Point().Mul(scalars[i], un(points[i]))}
		}
		return out, nil
	}
	sb, err := scalarBytes(scalars)
	if err != nil {
		return nil, err
	}
	pb, err := encodings(points, g.pointLen())
	if err != nil {
		return nil, err
	}
	out, st, err := g.mul(sb, pb)
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	return g.decode(out)
}

func (g *Group) Commit(coeffs []kyber.Scalar, base kyber.Point) ([]kyber.Point, error) {
	if len(coeffs) < MinDeviceBatch && !SingleOpOnDevice {
		out := make([]kyber.Point, len(coeffs))
		for i := range out {
			if base == nil {
				out[i] = &Point{g: g, p: g.inner.Point().Mul(coeffs[i], nil)}
			} else {
				out[i] = &Point{g: g, p: g.inner.Point().Mul(coeffs[i], un(base))}
			}
		}
		return out, nil
	}
	sb, err := scalarBytes(coeffs)
	if err != nil {
		return nil, err
	}
	var out, st []byte
	if base == nil {
		out, st, err = g.mulBase(sb)
	} else {
		var bb []byte
		if bb, err = un(base).MarshalBinary(); err != nil {
			return nil, err
		}
		out, st, err = g.mulSame(sb, bb)
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	return g.decode(out)
}

func (g *Group) MSM(scalars []kyber.Scalar, points []kyber.Point, bits uint) (kyber.Point, error) {
	if len(scalars) != len(points) {
		return nil, errLen
	}
	if len(scalars) < MinDeviceBatch && !SingleOpOnDevice { // the reference's own N x (Mul + Add) (share/poly.go:340-348)
		acc := g.inner.Point().Null()
		for i := range scalars {
			acc = acc.Add(acc, g.inner.Point().Mul(scalars[i], un(points[i])))
		}
		return &Point{g: g, p: acc}, nil
	}
	sb, err := scalarBytes(scalars)
	if err != nil {
		return nil, err
	}
	pb, err := encodings(points, g.pointLen())
	if err != nil {
		return nil, err
	}
	fl := g.trusted()
	if bits != 0 {
		fl |= hip.ScalarBits(bits)
	}
	var out, st []byte
	switch g.kind {
	case hip.Ed25519:
		out, st, err = hip.Ed25519MSM(sb, pb, fl)
	case hip.Bls12381G1:
		out, st, err = hip.Bls12381G1MSM(sb, pb, fl)
	case hip.Bls12381G2:
		out, st, err = hip.Bls12381G2MSM(sb, pb, fl)
	case hip.Bn256G1:
		out, st, err = hip.Bn256G1MSM(sb, pb, fl)
	case hip.Bn256G2:
		out, st, err = hip.Bn256G2MSM(sb, pb, fl)
	case hip.Bn254G1:
		out, st, err = hip.Bn254G1MSM(sb, pb, fl)
	case hip.Bn254G2:
		out, st, err = hip.Bn254G2MSM(sb, pb, fl)
	default:
		err = fmt.Errorf("kyberhip: %s has no device MSM", g.name)
	}
	if err != nil {
		return nil, err
	}
	if err = firstBad(st); err != nil {
		return nil, err
	}
	p := g.inner.Point()
	if err = p.UnmarshalBinary(out); err != nil {
		return nil, err
	}
	return &Point{g: g, p: p}, nil
}

func (g *Group) Validate(encs [][]byte) ([]byte, error) {
	size := g.pointLen()
	buf := make([]byte, 0, size*len(encs))
	bad := make([]int, 0)
	for i, e := range encs {
		if len(e) != size { // a wrong-length element fails alone, it must not shift its neighbours
			bad = append(bad, i)
			e = make([]byte, size)
		}
		buf = append(buf, e...)
	}
	var st []byte
	var err error
	switch g.kind {
	case hip.Ed25519:
		_, st, err = hip.Ed25519Unmarshal(buf)
	case hip.Bls12381G1:
		_, st, err = hip.Bls12381G1Unmarshal(buf, 0)
	case hip.Bls12381G2:
		_, st, err = hip.Bls12381G2Unmarshal(buf, 0)
	case hip.Bn256G1:
		_, st, err = hip.Bn256G1Unmarshal(buf)
	case hip.Bn256G2:
		_, st, err = hip.Bn256G2Unmarshal(buf)
	case hip.Bn254G1:
		_, st, err = hip.Bn254G1Unmarshal(buf, 0)
	case hip.Bn254G2:
		_, st, err = hip.Bn254G2Unmarshal(buf, 0)
	default:
		err = fmt.Errorf("kyberhip: %s has no device validation", g.name)
	}
	for _, i := range bad {
		if err == nil {
			st[i] = 1
		}
	}
	return st, err
}
