"""Static memory-instruction profile of a gfx950 assembly file (hipcc --cuda-device-only -S): per function, the
number of VALU / flat / scratch / global / LDS instructions and the dwords they move.  Used to see which out-of-line
functions pay for operands crossing call boundaries."""
import re
import sys
from collections import OrderedDict

W = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "b32": 1, "b64": 2, "b96": 3, "b128": 4, "ubyte": 1, "byte": 1, "short": 1, "ushort": 1}
fn = None
stats = OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
    if m and not m.group(1).startswith((".L", "BB")):
        fn = m.group(1)
        stats.setdefault(fn, dict(valu=0, mad=0, ld=0, st=0, ld_dw=0, st_dw=0, lds=0, salu=0, call=0, kinds={}))
        continue
    if fn is None:
        continue
    t = line.strip().split()
    if not t or t[0].startswith((";", ".", "//")):
        continue
    op = t[0]
    s = stats[fn]
    if op.startswith("v_"):
        s["valu"] += 1
        s["mad"] += op.startswith("v_mad_u64_u32")
    elif op.startswith(("flat_", "scratch_", "global_", "buffer_")):
        kind = op.split("_")[0]
        width = W.get(op.split("_")[-1], 1)
        if "_load" in op:
            s["ld"] += 1
            s["ld_dw"] += width
        elif "_store" in op:
            s["st"] += 1
            s["st_dw"] += width
        s["kinds"][kind] = s["kinds"].get(kind, 0) + 1
    elif op.startswith("ds_"):
        s["lds"] += 1
    elif op.startswith("s_swappc") or op.startswith("s_call"):
        s["call"] += 1
    elif op.startswith("s_"):
        s["salu"] += 1
rows = [(k, v) for k, v in stats.items() if v["valu"] + v["ld"] + v["st"] > 0]
rows.sort(key=lambda kv: -(kv[1]["ld_dw"] + kv[1]["st_dw"]))
print(f"{'function':70s} {'valu':>7s} {'mad':>6s} {'ld':>5s} {'ld_dw':>6s} {'st':>5s} {'st_dw':>6s} {'calls':>5s}  kinds")
for k, v in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k[:70]:70s} {v['valu']:7d} {v['mad']:6d} {v['ld']:5d} {v['ld_dw']:6d} {v['st']:5d} {v['st_dw']:6d} {v['call']:5d}  {v['kinds']}")
