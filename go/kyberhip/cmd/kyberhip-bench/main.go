// kyberhip-bench emits the engine's suites in the schema of the reference's benchmark collection
// (benchmark/README.md:63-80, benchmark/benchmark.go:22-90: results[module][instance] = {group | name, description,
// benchmarks: {type: {operation: testing.BenchmarkResult}}}) so that docs/benchmark-app can show them next to the CPU
// suites, plus a "batch" type the CPU suites do not have: one engine call over 2^k elements, reported per element.
//
// NOT COMPILED in the repository that ships it (no Go toolchain there).
//
//go:build hip

package main

import (
	"encoding/json"
	"fmt"
	"os"
	"testing"

	"go.dedis.ch/kyber/v4"
	"go.dedis.ch/kyber/v4/hip/suite"
	"go.dedis.ch/kyber/v4/util/test"
)

func groupResults(g kyber.Group) map[string]map[string]testing.BenchmarkResult {
	gb := test.NewGroupBench(g)
	res := map[string]map[string]testing.BenchmarkResult{"scalar": {}, "point": {}, "batch": {}}
	res["scalar"]["add"] = testing.Benchmark(func(b *testing.B) { gb.ScalarAdd(b.N) })
	res["scalar"]["mul"] = testing.Benchmark(func(b *testing.B) { gb.ScalarMul(b.N) })
	res["scalar"]["inv"] = testing.Benchmark(func(b *testing.B) { gb.ScalarInv(b.N) })
	res["point"]["add"] = testing.Benchmark(func(b *testing.B) { gb.PointAdd(b.N) })
	res["point"]["mul"] = testing.Benchmark(func(b *testing.B) { gb.PointMul(b.N) })
	res["point"]["baseMul"] = testing.Benchmark(func(b *testing.B) { gb.PointBaseMul(b.N) })
	res["point"]["encode"] = testing.Benchmark(func(b *testing.B) { gb.PointEncode(b.N) })
	res["point"]["decode"] = testing.Benchmark(func(b *testing.B) { gb.PointDecode(b.N) })
	if bg, ok := g.(suite.BatchGroup); ok {
		const n = 1 << 16
		rnd := g.(kyber.Random).RandomStream()
		scalars := make([]kyber.Scalar, n)
		points := make([]kyber.Point, n)
		base, err := bg.Commit([]kyber.Scalar{g.Scalar().Pick(rnd)}, nil)
		if err != nil {
			panic(err)
		}
		for i := range scalars {
			scalars[i] = g.Scalar().Pick(rnd)
			points[i] = base[0]
		}
		// N = elements processed, T = wall time: ns/op in the front end is then time per element
		res["batch"]["mul"] = testing.Benchmark(func(b *testing.B) {
			for done := 0; done < b.N; done += n {
				if _, err := bg.BatchMul(scalars, points); err != nil {
					b.Fatal(err)
				}
			}
		})
		res["batch"]["baseMul"] = testing.Benchmark(func(b *testing.B) {
			for done := 0; done < b.N; done += n {
				if _, err := bg.Commit(scalars, nil); err != nil {
					b.Fatal(err)
				}
			}
		})
		res["batch"]["msm"] = testing.Benchmark(func(b *testing.B) {
			for done := 0; done < b.N; done += n {
				if _, err := bg.MSM(scalars, points, 0); err != nil {
					b.Fatal(err)
				}
			}
		})
	}
	return res
}

func main() {
	out := "data.hip.json"
	if len(os.Args) > 1 {
		out = os.Args[1]
	}
	results := map[string]map[string]map[string]any{"groups": {}}
	for _, g := range []kyber.Group{suite.NewBlakeSHA256Ed25519HIP(), suite.NewGroupSuiteBLS12381(), suite.NewGroupSuiteBn256(),
		suite.NewGroupSuiteBn254()} {
		fmt.Printf("Running benchmarks for group %s...\n", g.String())
		results["groups"][g.String()] = map[string]any{
			"group":       g.String(),
			"description": "MI355X batch engine (libkyberhip) behind the kyber interfaces; variable-time",
			"benchmarks":  groupResults(g),
		}
	}
	f, err := os.Create(out)
	if err != nil {
		fmt.Println("Error creating output file:", err)
		return
	}
	defer f.Close()
	enc := json.NewEncoder(f)
	enc.SetIndent("", "  ")
	if err := enc.Encode(results); err != nil {
		fmt.Println("Error encoding JSON:", err)
	}
}
