#!/bin/bash
# Round 2, call 1: re-profile the binaries at round-1 HEAD (new Ed25519 carry chain, packed Montgomery tail), which
# round 1 could only count statically.  Parity first, then trace + separate PMC passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_head; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other --no-host-path"
timeout 300 $B > $O/bench_ed.json 2> $O/bench_ed.err; cat $O/bench_ed.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ed_trace -- $B > $O/ed_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ed_fetch -- $B > $O/ed_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ed_write -- $B > $O/ed_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o ed_sq -- $B > $O/ed_sq.log 2>&1
for s in bls12381 bn256; do timeout 200 python tools/pair_probe.py $s 65536 2>/dev/null | tail -1 | tee $O/probe_$s.json; done
P="python tools/pair_probe.py bls12381 65536"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o bls12381_fetch -- $P > $O/bls12381_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o bls12381_write -- $P > $O/bls12381_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o bls12381_sq -- $P > $O/bls12381_sq.log 2>&1
timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee $O/msm_probe_2p20.json
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
ls $O
