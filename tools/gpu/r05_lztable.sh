#!/bin/bash
# Round 5: the ladders' window table built in the lazy form (jaclz_table8: 1 doubling + 6 mixed additions, one shared inversion)
# against the packed builder (libkyberhip_pktab.so = the previous commit's units), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_lztable; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_bls12381.py tests/test_gpu_switches.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
for rep in 1 2; do for lib in "" kyber_amd/lib/libkyberhip_pktab.so; do
  tag="{\"lib\": \"${lib:-lazy table}\", "
  KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py bn256 262144 7 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['lib'][-12:], d['suite'], {k: round(v, 2) for k, v in d.items() if k.endswith('_ms')})"
  KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py bls12381 65536 7 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['lib'][-12:], d['suite'], {k: round(v, 2) for k, v in d.items() if k.startswith('g1') and k.endswith('_ms')})"
done; done
