#!/usr/bin/env python3
"""Extract the bn254 known answers and constants held by the reference into tests/golden/bn254.json.
Run in the build container only (needs /root/reference).

Sources (pairing/bn254):
  test_vectors_test.go:5-511      hashToFieldTestVectors: 100 x (msg, x, y), DST "BLS_SIG_BN254G1_XMD:KECCAK-256_SSWU_RO_NUL_"
  test_vectors_test.go:512-5519   mapToPointTestVectors: 1000 x (u, x, y) of the Shallue-van de Woestijne map
  point_test.go:14-48             two pointG1.Hash outputs, DST "domain_separation_tag_test_12345"
  point_test.go:50-79             one expand_message_xmd (Keccak-256) output
  constants.go:72-84              the four SvdW constants (Montgomery form -> plain)
  curve.go:19-23, twist.go:16-33  generators and the twist's b (Montgomery form -> plain)
"""
import json
import os
import re

REF = "/root/reference/pairing/bn254"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254.json")
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
RINV = pow(1 << 256, -1, P)


def plain(words):
    """gfP{w0, w1, w2, w3} (little-endian 64-bit words, Montgomery form) -> integer"""
    return sum(int(w, 16) << (64 * i) for i, w in enumerate(words)) * RINV % P


def gfps(text):
    return [plain([w.strip() for w in m.group(1).split(",")])
            for m in re.finditer(r"gfP\{(0x[0-9a-f]+, 0x[0-9a-f]+, 0x[0-9a-f]+, 0x[0-9a-f]+)\}", text)]


vec = open(os.path.join(REF, "test_vectors_test.go")).read()
h2f = re.findall(r'Msg:\s+"([0-9a-f]*)",\s+RefX:\s+"([0-9a-f]{64})",\s+RefY:\s+"([0-9a-f]{64})"', vec)
m2p = re.findall(r'U:\s+"(\d+)",\s+RefX:\s+"(\d+)",\s+RefY:\s+"(\d+)"', vec)
assert len(h2f) == 100 and len(m2p) == 1000, (len(h2f), len(m2p))
pt = open(os.path.join(REF, "point_test.go")).read()
dom = re.search(r'domain := \[\]byte\("([^"]+)"\)', pt).group(1)
h1 = re.search(r'Hash\(\[\]byte\("([^"]+)"\)\).*?DecodeString\("([0-9a-f]{128})"\)', pt, re.S)
h2 = re.search(r'buf2, err := hex\.DecodeString\("([0-9a-f]{64})"\).*?refBuf2, err := hex\.DecodeString\("([0-9a-f]{128})"\)', pt, re.S)
ex = re.search(r'dst := \[\]byte\("([^"]+)"\)\s+msg, err := hex\.DecodeString\("([0-9a-f]+)"\).*?EncodeToString\(expanded\) != "([0-9a-f]{192})"', pt, re.S)
cs = open(os.path.join(REF, "constants.go")).read()
svdw = {}
for name in ("c1", "c2", "c3", "c4"):
    svdw[name] = str(gfps(re.search(r"var %s = &gfP\{[^}]*\}" % name, cs).group(0))[0])
tw = gfps(open(os.path.join(REF, "twist.go")).read())  # twistB x, y; twistGen x.x x.y y.x y.y
assert len(tw) == 6
cv = open(os.path.join(REF, "curve.go")).read()
gen = re.search(r"var curveGen = &curvePoint\{\s*x: \*newGFp\((-?\d+)\),\s*y: \*newGFp\((-?\d+)\)", cv)
json.dump({"h2f_dst": "BLS_SIG_BN254G1_XMD:KECCAK-256_SSWU_RO_NUL_",
           "hash_to_field": [{"msg": m, "x": x, "y": y} for m, x, y in h2f],
           "map_to_point": [[u, x, y] for u, x, y in m2p],
           "hash_g1_dst": dom,
           "hash_g1": [{"msg_hex": h1.group(1).encode().hex(), "point": h1.group(2)}, {"msg_hex": h2.group(1), "point": h2.group(2)}],
           "expand": {"dst": ex.group(1), "msg_hex": ex.group(2), "out": ex.group(3)},
           "svdw": svdw,
           "twist_b": [str(tw[0]), str(tw[1])],            # gfP2{x, y} = x i + y
           "twist_gen": [str(v) for v in tw[2:6]],         # x.x, x.y, y.x, y.y
           "curve_gen": [gen.group(1), gen.group(2)]}, open(OUT, "w"), indent=0)
print("ok", OUT, os.path.getsize(OUT))
