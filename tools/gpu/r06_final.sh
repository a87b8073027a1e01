#!/bin/bash
# Round-6 evidence on the shipped tree: all GPU tests, smoke, then for every kernel a roofline object quotes a kernel trace
# + three separate PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE) next to the SHA-256 of the sources it was built from
# (tools/source_digest.py: tools/roofline_inputs.py refuses a profile whose kernel's sources changed since), the default
# bench line as the driver runs it (compact line + bench_detail.json), the forced-RCCL world-1 line, the data.json-schema
# figures.  Outputs under gpurun_out/r06_final; copy the .txt / .json summaries to profiles/r06_final_* and run
# `python tools/roofline_inputs.py profiles r06_final`.   usage: r06_final.sh [tests|prof|bench]... (default: all three)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_final; mkdir -p $O; export TMPDIR=/tmp
WHAT="${*:-tests prof bench}"
if [[ " $WHAT " == *" tests "* ]]; then
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
prof() {  # prof <name> <command...>: trace + three PMC passes + the digest of the sources
  n=$1; shift
  python tools/source_digest.py $O/${n}_meta.json
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ${n}_trace -- "$@" > $O/${n}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ${n}_fetch -- "$@" > $O/${n}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ${n}_write -- "$@" > $O/${n}_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o ${n}_sq -- "$@" > $O/${n}_sq.log 2>&1
}
if [[ " $WHAT " == *" prof "* ]]; then
prof ed python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other --no-host-path
prof bls12381 python tools/pair_probe.py bls12381 65536
prof verify python tools/verify_probe.py 65536
prof gtmul python tools/gt_probe.py 65536 3
prof bn256 python tools/pair_probe.py bn256 262144
prof bn254 python tools/pair_probe.py bn254 262144
prof mul python tools/mul_probe.py bls12381 65536 3
KYB_LVM_MIN=1000000000 prof mulperlane python tools/mul_probe.py bls12381 65536 3
prof msm_bls python tools/msm_bls_probe.py 1048576 5 all
prof msm_affine python tools/msm_bls_probe.py 1048576 20 affine   # the trusted uncompressed convention alone: the per-stage times of the 4.4 ms figure
prof mulbn256 python tools/mul_probe.py bn256 262144 3
prof fb python tools/commit_probe.py 1048576   # bench.py's own g1_commit call and inputs
for s in "bls12381 65536" "bn256 262144" "bn254 262144"; do set -- $s; timeout 300 python tools/pair_probe.py $1 $2 2>/dev/null | tail -1 > $O/probe_$1.json; done
timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 > $O/mul_probe_65536.json
timeout 300 python tools/mul_probe.py bn256 262144 2>/dev/null | tail -1 > $O/mul_probe_bn256.json
timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 > $O/verify_probe.json
timeout 300 python tools/commit_probe.py 1048576 2>/dev/null | tail -1 > $O/commit_probe.json
timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 > $O/fb_probe_bls12381.json
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 > $O/msm_probe_2p20.json
timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1 > $O/single_call_latency.json
timeout 300 python tools/ed_probe.py 2>/dev/null | tail -1 > $O/ed_probe.json
timeout 300 python tools/keyline_probe.py 64 24 2>/dev/null | tail -1 > $O/keyline_probe.json
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
timeout 600 python tools/data_json.py $O/data.hip.json > $O/data_json.log 2>&1; tail -1 $O/data_json.log
fi
if [[ " $WHAT " == *" bench "* ]]; then
KYB_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_world1_forced_dist.json 2> $O/bench_world1.err
cp bench_detail.json $O/bench_world1_forced_dist_detail.json 2>/dev/null
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python3 -c "
import json
s = open('$O/bench.json').read().strip().split('\n'); print('stdout lines', len(s), 'bytes', len(s[-1])); print(s[-1][:1500])"
fi
