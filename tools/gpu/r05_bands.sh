#!/bin/bash
# Round 5: which G1 path above 2^16 now that the per-lane ladder runs on lazy limbs: the lane machine (default) or the per-lane kernel (KYB_LVM_MIN huge)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_bands; mkdir -p $O
for n in 131072 262144 1048576; do for lv in default 1000000000; do
  if [ $lv = default ]; then unset KYB_LVM_MIN; else export KYB_LVM_MIN=$lv; fi
  timeout 300 python tools/mul_probe.py bls12381 $n 5 2>/dev/null | tail -1 | tee -a $O/bands.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['n'], d['lvm_min'], {k: round(v, 2) for k, v in d.items() if k.startswith('g1') and k.endswith('_ms')})"
done; done
