#!/bin/bash
# Round 5, first look: the compact bench line as the driver runs it (size + parse), and per-stage traces of the
# BLS12-381 G1 MSM and the bn256 Mul kernels on this box before any change (the A of this round's A/Bs).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_first; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python3 - <<P
import json
s = open("$O/bench.json").read(); lines = s.strip().split("\n")
print("stdout lines", len(lines), "bytes of last", len(lines[-1])); d = json.loads(lines[-1]); print(json.dumps(d)[:3000])
P
cp bench_detail.json $O/ 2>/dev/null
timeout 300 python tools/msm_bls_probe.py 1048576 15 all 2>/dev/null | tail -1 | tee -a $O/msm_ab.jsonl
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o msm_trace -- python tools/msm_bls_probe.py 1048576 10 > $O/msm_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -30 $O/msm_trace.txt | cut -c1-160
