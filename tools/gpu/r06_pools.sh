#!/bin/bash
# round 6: staging pools (two host threads on one device overlap) -- the new concurrency tests, the older concurrency / device
# tests and the same-key cache test under the new layout, and the two-thread figure with one pool (KYB_STAGE_POOLS=1) beside it
set -u
O=gpurun_out/r06_pools; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_verify_same_key.py tests/test_gpu_devices.py "tests/test_gpu_soak.py::test_concurrent_host_threads_and_streams" "tests/test_gpu_soak.py::test_concurrent_round4_entry_points" tests/test_gpu_ed25519.py tests/test_gpu_fixed_base.py -x -q -s > $O/tests.log 2>&1; tail -6 $O/tests.log; grep "one call" $O/tests.log
KYB_STAGE_POOLS=1 timeout 600 python -m pytest tests/test_gpu_concurrency.py::test_two_host_threads_overlap_on_one_device -x -q -s > $O/one_pool.log 2>&1; grep "one call" $O/one_pool.log; tail -2 $O/one_pool.log
