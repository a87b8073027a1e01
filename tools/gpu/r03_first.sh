#!/bin/bash
# Round 3, first GPU call: the new parity tests (IBE GT-bytes pin, MSM at 2^20, off-subgroup bn256 G2 base) and
# baseline probes of the G1 / G2 multiplication kernels at three batch sizes (how much of the 2^16 figure is latency).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_first; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py::test_ibe_vector_pins_pair_bytes_on_the_engine \
  tests/test_gpu_bn256.py::test_same_base_commit_with_an_off_subgroup_g2_base_at_2p18 \
  tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation -x -q > $O/pytest_new.log 2>&1
echo "rc=$?" >> $O/pytest_new.log; tail -15 $O/pytest_new.log
for n in 65536 262144 1048576; do
  timeout 300 python tools/pair_probe.py bls12381 $n 2>/dev/null | tail -1 | tee $O/probe_bls12381_$n.json
done
timeout 300 python tools/pair_probe.py bn256 262144 2>/dev/null | tail -1 | tee $O/probe_bn256_262144.json
