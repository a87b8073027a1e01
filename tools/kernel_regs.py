#!/usr/bin/env python3
"""Register / scratch / LDS allocation of the kernels of one built object (code-object metadata; no GPU needed).
usage: tools/kernel_regs.py kyber_amd/csrc/bls12381.o [substring]"""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, os.path.basename(obj)); shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = glob.glob(os.path.join(tmp, "*gfx950*"))[0]
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
for blk in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sub not in name: continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    print(f"{name[:110]:110s} vgpr {g('vgpr_count'):4d} agpr {int(blk.split()[0]):4d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):6d} lds {g('group_segment_fixed_size'):6d}")
