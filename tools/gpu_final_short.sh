#!/bin/bash
# Short round-end evidence (when GPU minutes are scarce): GPU tests, smoke, bench, the pairing kernel's HBM traffic, the
# lone-wave probes.  tools/gpu_final.sh is the full version.
mkdir -p gpurun_out/final; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_gpu.log; tail -3 gpurun_out/final/pytest_gpu.log
timeout 120 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; cat gpurun_out/final/bench.json
P="python tools/pair_probe.py bls12381 65536"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final -o bls12381_fetch -- $P > gpurun_out/final/bls12381_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final -o bls12381_write -- $P > gpurun_out/final/bls12381_write.log 2>&1
timeout 60 $P 2>/dev/null | tail -1 > gpurun_out/final/probe_bls12381_2p16.json
(timeout 60 ./tools/tower_probe.bin 1024 0; timeout 60 ./tools/tower_probe.bin 8 0) > gpurun_out/final/tower_probe.json 2>&1
timeout 60 ./tools/fpmul_probe.bin > gpurun_out/final/fpmul_probe.json 2>&1
for f in gpurun_out/final/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep "pair_kernel |" gpurun_out/final/bls12381_fetch.txt gpurun_out/final/bls12381_write.txt | grep -v "kernel | [0-9]"
