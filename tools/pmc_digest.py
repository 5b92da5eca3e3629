#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc CSV (short kernel names), one line per (kernel, template args)."""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(gemm_\w+|attn_\w+|\w+_kernel)(<[^>]*>)?", k)
    k = (m.group(0) if m else k[:40]) + f" grid{r['Grid_Size']}"
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in acc.items():
    if "gemm" not in k and "attn" not in k: continue
    print(k)
    for c, x in sorted(v.items()):
        cnt = n[(k, c)]
        print(f"    {c:40s} {x / cnt:14.4g} per launch  ({cnt} launches, {dur[k] / sum(1 for q in n if q[0] == k) / cnt * len(v):.1f} us)")
