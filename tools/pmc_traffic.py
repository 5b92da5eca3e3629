#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in KiB... units per guide:
FETCH_SIZE/WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section) -> doubled here).  usage: pmc_traffic.py fetch.csv write.csv"""
import collections
import csv
import sys


BY_GRID = "--by-grid" in sys.argv          # one row per (kernel, grid): the launches of one kernel differ by orders of magnitude in a training step
TOP = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14


def load(path, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:48] if "anonymous" in k else k[:48]
        if BY_GRID:
            k += " grid=" + str(r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


def symbol(raw):
    """'void (anonymous namespace)::attn_bf16_kernel<true, 1, 8, false>(unsigned short const*, ...' -> the template id"""
    k = raw
    if "::" in k and "anonymous" in k:
        k = k[k.find("::") + 2:]
    depth = 0
    for i, ch in enumerate(k):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return k[:i]
    return k


def load_full(path, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        acc[symbol(r["Kernel_Name"])][0] += 1
        acc[symbol(r["Kernel_Name"])][1] += float(r["Counter_Value"])
    return acc


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    if "--json" in sys.argv:        # per kernel SYMBOL (what bench.py's roofline rows are keyed by) + the two classes -> profiles/rNN_traffic.json
        import json
        ff, wf = load_full(sys.argv[1], "FETCH_SIZE"), load_full(sys.argv[2], "WRITE_SIZE")
        out = {"note": "HBM bytes per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes over `bench.py --eager "
                       "--steps 2 --warmup 1`); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); "
                       "KB -> bytes x1024"}
        for cls, pat in (("gemm_bf16", "gemm_"), ("attention_bf16", "attn_bf16")):
            nf = sum(v[0] for k, v in f.items() if pat in k)
            fb = sum(v[1] for k, v in f.items() if pat in k) * 2.0 * 1024
            nw = sum(v[0] for k, v in w.items() if pat in k)
            wb = sum(v[1] for k, v in w.items() if pat in k) * 1024
            if nf and nw:
                out[cls] = {"launches": nf, "fetch_bytes_per_launch": round(fb / nf), "write_bytes_per_launch": round(wb / nw),
                            "traffic_bytes_per_launch": round(fb / nf + wb / nw)}
        for k in ff:
            if k in wf and (k.startswith("gemm_") or k.startswith("attn_")):
                fb, wb = ff[k][1] * 2.0 * 1024 / ff[k][0], wf[k][1] * 1024 / wf[k][0]
                out[k] = {"launches": ff[k][0], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                          "traffic_bytes_per_launch": round(fb + wb)}
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("| kernel | launches | fetch MB/launch (x2 corrected) | write MB/launch | total MB/launch |")
    print("|---|---|---|---|---|")
    rows = []
    for k in f:
        n = f[k][0]
        fm = 2.0 * f[k][1] * 1024 / n / 1e6
        wm = w.get(k, [1, 0.0])[1] * 1024 / max(w.get(k, [1, 0.0])[0], 1) / 1e6
        rows.append((n * (fm + wm), k, n, fm, wm))
    for _, k, n, fm, wm in sorted(rows, reverse=True)[:TOP]:
        print(f"| {k} | {n} | {fm:.2f} | {wm:.2f} | {fm + wm:.2f} |")


if __name__ == "__main__":
    main()
