#!/bin/bash
# BLS12-381 fixed-base kernels in their own two-wave translation unit (bls12381_fb.hip): tests, then fb_probe
cd /root/repo; mkdir -p gpurun_out/r04_fbtu; O=gpurun_out/r04_fbtu
timeout 1200 python -m pytest tests/test_gpu_fixed_base.py tests/test_gpu_bls12381.py tests/test_gpu_callers.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_switches.py -q -x -k "fb" > $O/switch.log 2>&1; tail -2 $O/switch.log
timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | tee -a $O/fb.jsonl
timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | tee -a $O/fb.jsonl
