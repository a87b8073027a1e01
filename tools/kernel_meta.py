"""Print register / LDS / scratch allocations of every kernel in a built object (code-object metadata, no GPU)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
for obj in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = glob.glob(os.path.join(tmp, "*gfx950*"))[0]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        print(f"{g('name')[:60]:60s} vgpr={g('vgpr_count'):>4s} lds={g('group_segment_fixed_size'):>6s} "
              f"scratch={g('private_segment_fixed_size'):>6s} spill={g('vgpr_spill_count'):>4s}")
