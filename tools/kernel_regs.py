#!/usr/bin/env python3
"""Per-kernel resource table from a hipcc -save-temps .s file: VGPR / AGPR / SGPR counts, spills, scratch, LDS.
usage: tools/kernel_regs.py file.s [substring]"""
import re, sys, subprocess
text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if flt in name:
        print(f"{name[:70]:70s} agpr {blk.split()[0]:>4s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} "
              f"vspill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size')}")
