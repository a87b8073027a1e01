#!/bin/bash
# Same-box A/B of library builds on the Ed25519 kernels + the per-instruction issue costs (tools/valu_rates.hip)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_ed_ab; mkdir -p $O; rm -f $O/*.jsonl
[ -n "$SKIP_RATES" ] || { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && /tmp/valu_rates | tee $O/valu_rates.jsonl; }
for round in 1 2; do for v in ${VARIANTS:-A S C D}; do
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/ed_probe.py 2>>$O/err.log | tail -1 | sed "s/^{/{\"v\": \"$v\", /" | tee -a $O/ab.jsonl
done; done
