#!/bin/bash
# After tools/gpu/r06_final.sh (+ r06_soak.sh) ran on the GPU box: copy the summaries the judge reads from gpurun_out/ (scratch)
# into profiles/ (tracked) and regenerate profiles/roofline_inputs.json (which refuses a profile whose kernel's sources changed).
# usage: tools/collect_final.sh [round-tag]   (default r06_final)
cd "$(dirname "$0")/.."
R=${1:-r06_final}; O=gpurun_out/$R
for f in $O/*_trace.txt $O/*_fetch.txt $O/*_write.txt $O/*_sq.txt $O/*_meta.json $O/*probe*.json $O/single_call_latency.json \
         $O/data.hip.json $O/pytest_gpu.log $O/bench.json $O/bench_detail.json $O/bench_world1_forced_dist.json \
         $O/bench_world1_forced_dist_detail.json; do
  [ -s "$f" ] && cp "$f" profiles/${R}_$(basename "$f")
done
python tools/roofline_inputs.py profiles $R
