#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 65536 131072 262144 524288 1048576; do
timeout 300 python bench.py --n $n --steps 5 --warmup 2 --no-cpu-baseline --no-other --no-host-path 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['elements_per_gpu'], 'var ms %.3f'%d['detail']['var_base_kernel_ms'], 'fix ms %.3f'%d['detail']['fixed_base_kernel_ms'])"
done
