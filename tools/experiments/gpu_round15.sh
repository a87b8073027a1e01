#!/bin/bash
set -x
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak 2>/dev/null && timeout 120 /tmp/valu_peak > gpurun_out/valu_peak2.json 2>&1
cat gpurun_out/valu_peak2.json
