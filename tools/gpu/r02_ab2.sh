#!/bin/bash
# Same-box A/B of two builds of the library (tools/ab_build.sh A ..., B ...): the three pairing probes, alternating,
# three rounds; the medians go to gpurun_out/r02_ab2/summary.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_ab2; mkdir -p $O; rm -f $O/*.jsonl
for round in 1 2 3; do
for v in A B; do
for s in "bls12381 65536" "bn256 262144" "bn254 262144"; do set -- $s
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/pair_probe.py $1 $2 2>/dev/null | tail -1 >> $O/$v.jsonl
done; done; done
python - <<'PY'
import json, statistics
out = {}
for v in "AB":
    rows = [json.loads(l) for l in open("gpurun_out/r02_ab2/%s.jsonl" % v)]
    for s in ("bls12381", "bn256", "bn254"):
        r = [x for x in rows if x["suite"] == s]
        out.setdefault(s, {})[v] = {k: round(statistics.median(x[k] for x in r), 3) for k in ("pair_ms", "pair_validated_ms", "pair_check_ms")}
json.dump(out, open("gpurun_out/r02_ab2/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
