#!/bin/bash
# tower machine: parity (pairing-suite GPU tests incl. the config-size KATs), then throughput of the pairing entry points
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_tvm; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_bn256.py tests/test_gpu_full_size.py tests/test_gpu_callers.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for s in bls12381 bn256; do timeout 300 python tools/pair_probe.py $s 65536 2>$O/probe.err | tail -1 | tee $O/probe_$s.json; done
timeout 300 python tools/pair_probe.py bn256 262144 2>$O/probe.err | tail -1 | tee $O/probe_bn256_2p18.json
