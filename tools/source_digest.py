#!/usr/bin/env python3
"""SHA-256 of every source file of libkyberhip.so (kyber_amd/csrc, include/), as JSON {file: digest}.

The profiling scripts (tools/gpu/r04_*.sh) write this next to every trace / PMC summary they produce -- on the GPU
box, from the tree the profiled binary was built from -- and tools/roofline_inputs.py refuses a profile whose kernel's
sources have changed since (no roofline object may quote a profile of another binary)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digests():
    out = {}
    for d in ("kyber_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".hip", ".cuh", ".h", ".py", ".inc")) and not f.startswith(("tower_vm_b", "lane_vm_b")):
                out[f"{d}/{f}"] = hashlib.sha256(open(os.path.join(ROOT, d, f), "rb").read()).hexdigest()
    return out


if __name__ == "__main__":
    json.dump({"sources": digests()}, open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout, indent=0)
