#!/bin/bash
# A/B of a tower-machine change: parity (known answers inside config-size batches + suite tests), then the three probes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bn254.py tests/test_gpu_bn256.py tests/test_gpu_bls12381.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for s in "bls12381 65536" "bn256 262144" "bn254 262144"; do set -- $s; timeout 300 python tools/pair_probe.py $1 $2 2>/dev/null | tail -1 | tee $O/probe_$1.json; done
