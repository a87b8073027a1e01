#!/bin/bash
# round 6: every environment switch (the new ones included) against the oracles, the 30-seed randomised soak on the final
# binaries, and the default bench line once more with the regenerated profiles/roofline_inputs.json
set -u
O=gpurun_out/r06_soak; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_switches.py -q > $O/switches.log 2>&1; tail -3 $O/switches.log
KYB_SOAK_SEEDS=30 timeout 2400 python -m pytest tests/test_gpu_soak.py -q > $O/soak.log 2>&1; tail -3 $O/soak.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python3 -c "
import json
s = open('$O/bench.json').read().strip().split('\n'); print('stdout lines', len(s), 'bytes', len(s[-1])); d = json.loads(s[-1]); print(d['roofline']); print(d['checks'])"
