"""Opcode histogram of the loops of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S):
python tools/isa_loop_hist.py file.s <kernel-name-substring> [min_valu].  Used to count the instructions of the
Ed25519 window loop (VALU-bound at 99 % busy: time follows the count)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
start = next(i for i, l in enumerate(lines) if l.startswith("_") and name in l and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
for i, l in enumerate(body):
    t = l.strip().split()
    if not t or not (t[0].startswith("s_cbranch") or t[0] == "s_branch"):
        continue
    tgt = t[-1]
    if tgt not in labels or labels[tgt] >= i:
        continue
    ops = collections.Counter()
    for x in body[labels[tgt]:i]:
        tt = x.strip().split()
        if tt and not tt[0].startswith((".", ";")) and not tt[0].endswith(":"):
            ops[re.sub(r"_e32|_e64", "", tt[0])] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    if valu < min_valu:
        continue
    print(f"loop {tgt}: {valu} VALU, {sum(v for k, v in ops.items() if k.startswith('v_mad_'))} MAD")
    print("   " + ", ".join(f"{k} {v}" for k, v in ops.most_common(16)))
