#!/bin/bash
# re-take the traces whose per-kernel averages must match the bench line (no host-path launches mixed in)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final; export TMPDIR=/tmp
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-path"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final -o bench_trace -- $B > gpurun_out/final/bench_trace.log 2>&1
B2="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other --no-host-path"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final -o ed_fetch -- $B2 > gpurun_out/final/ed_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final -o ed_write -- $B2 > gpurun_out/final/ed_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/final -o ed_sq -- $B2 > gpurun_out/final/ed_sq.log 2>&1
for f in gpurun_out/final/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep "ed25519" gpurun_out/final/bench_trace.txt | head; grep "ed25519" gpurun_out/final/ed_fetch.txt gpurun_out/final/ed_write.txt | head
