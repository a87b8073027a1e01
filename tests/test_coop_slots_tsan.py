"""The slot schedule of the MSM's cooperative additions / doublings (kyber_amd/csrc/coop_slots.cuh) under
ThreadSanitizer: the host harness runs the four lanes of a group as four threads with a pthread barrier for the
workgroup barrier, so a slot that one lane writes while another still reads it within a level -- on the GPU a sum that is
wrong now and then -- is a data race the sanitizer names.  Builds its own binary (about a minute)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_slot_schedule_has_no_data_race(tmp_path):
    exe = str(tmp_path / "coop_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe,
                            os.path.join(HERE, "host_harness.cpp"), os.path.join(HERE, "coop_tsan_main.cpp")],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("ThreadSanitizer not available: " + build.stderr[:200])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in run.stderr and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot run in this container: " + run.stderr[:200])
    assert run.returncode == 0, run.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.stdout.count("st 0") == 7, run.stdout
