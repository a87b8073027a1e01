#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
KYB_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/dist_bench.json 2> gpurun_out/dist_bench.err
echo rc=$?
tail -3 gpurun_out/dist_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/dist_bench.json"))
print(d["value"], d["config"], json.dumps(d["other_workloads"]["bls12381_g1_msm_2p20"]), d["other_workloads"]["ed25519_msm_2p20"])
PY
