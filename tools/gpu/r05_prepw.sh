#!/bin/bash
# Round 5: the operand kernel's register budget again (1 wave = 512 registers, default; 2 waves) now that G1 decode and hash are shorter
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_prepw; mkdir -p $O
for rep in 1 2; do for lib in "" kyber_amd/lib/libkyberhip_prep2.so; do
  tag="{\"operand_waves\": \"${lib:-1}\", "
  KYBER_HIP_LIB=$lib timeout 200 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['operand_waves'][-10:], {k: round(v, 2) for k, v in d.items() if k.startswith('pair') and k.endswith('_ms')})"
  KYBER_HIP_LIB=$lib timeout 200 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['operand_waves'][-10:], {k: round(v, 2) for k, v in d.items() if k.endswith('_ms')})"
done; done
