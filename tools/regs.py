#!/usr/bin/env python3
"""Register / spill report of a device-only assembly listing (hipcc --cuda-device-only -S):
    python tools/regs.py /tmp/k256m.s [substring filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for b in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
    rows.append((re.search(r"\.name:\s+(\S+)", b).group(1), g("vgpr_count"), int(b.split("\n")[0]),
                 g("vgpr_spill_count"), g("sgpr_spill_count")))
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (n, vg, ag, sp, ss), d in zip(rows, dem):
    m = re.search(r"<(.*)>", d)
    t = (m.group(1) if m else d).replace("vptq::", "")
    if flt in d:
        print(f"{t:48s} vgpr={vg:4d} agpr={ag:3d} vspill={sp:3d} sspill={ss}")
