#!/usr/bin/env python3
"""Matrix-pipe and VALU utilisation per kernel symbol from one rocprofv3 PMC pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)  (the gfx94x derived-metric formula;
ROCm 7.2 ships no gfx950 section; rocprofv3 sums a counter over its instances, and GRBM_GUI_ACTIVE has one per XCD, hence / 8:
with that the global attention reads 48 % busy at 0.39 of the 2.4 GHz peak x 2.4 / 1.96 GHz under the profiler = 0.48).  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count quad-cycles
(MI355X_MICROARCH.md, per-instruction constants).  usage: pmc_util.py counter_collection.csv"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import symbol  # noqa: E402

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = symbol(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
print("| kernel | launches | GUI_ACTIVE cycles / launch (per XCD) | MFMA busy % | VALU active % of wave cycles | issue-stall % of wave cycles |")
print("|---|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0           # one instance per XCD, summed by rocprofv3
    if gui <= 0:
        continue
    mf = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4)
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rows.append((gui, k, cnt[k], gui / max(cnt[k], 1), mf, 100.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc))
for gui, k, n, per, mf, va, st in sorted(rows, reverse=True)[:16]:
    print(f"| {k[:64]} | {n} | {per:.0f} | {mf:.1f} | {va:.1f} | {st:.1f} |")
