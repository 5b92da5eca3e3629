#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 rocpd sqlite database: for the LAST `n` back-to-back repetitions of the
clip (a repetition starts at every patchify_kernel), after dropping the last `skip` ones (bench.py ends with `--steps`
eager instrumented repetitions; the graph replays come before them), prints span, busy time (union of the kernel
intervals), the idle remainder and the gap histogram.  Usage: tools/rocpd_gaps.py results.db [n] [skip] [marker]
(marker: the kernel that opens a repetition; default patchify_kernel = a clip; adamw_flat_kernel = a training step)"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select start, end, {name} from kernels order by start").fetchall()
    marker = sys.argv[4] if len(sys.argv) > 4 else "patchify_kernel"
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < n + skip + 1:
        print("not enough repetitions")
        return
    lo, hi = starts[-n - skip - 1], starts[-skip - 1]
    seg = rows[lo:hi]
    span = seg[-1][1] - seg[0][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, _))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"{n} clips, {len(seg)} kernels: span {span / 1e6 / n:.3f} ms per clip, busy {busy / 1e6 / n:.3f} ms, idle {(span - busy) / 1e6 / n:.3f} ms "
          f"({100 * (span - busy) / span:.1f} %), {len(gaps) / n:.0f} gaps per clip, mean gap {sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3:.2f} us")
    hist = {}
    for g, _ in gaps:
        b = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else ">=8us"
        hist[b] = hist.get(b, 0) + 1
    print("gap histogram per clip:", {k: round(v / n, 1) for k, v in hist.items()})
    # where the idle time sits: the repetition cut into 40 equal time slices, idle microseconds per slice
    t0, per = seg[0][0], span / 40.0
    slices = [0.0] * 40
    cur_e = seg[0][1]
    for s_, e_, _k in seg[1:]:
        if s_ > cur_e:
            slices[min(39, int((cur_e - t0) / per))] += (s_ - cur_e) / 1e3 / n
        cur_e = max(cur_e, e_)
    print("idle us per 1/40 of the repetition(s):", " ".join(f"{x:.0f}" for x in slices))
    byk = {}
    for g, k in gaps:
        kk = k.split("(")[0][-60:]
        byk[kk] = byk.get(kk, 0.0) + g / 1e3 / n
    print("idle us per repetition by the kernel that follows the gap:")
    for kk, v in sorted(byk.items(), key=lambda t: -t[1])[:14]:
        print(f"  {v:8.1f}  {kk}")
    big = sorted(gaps, key=lambda t: -t[0])[:12]
    print("largest gaps (us, kernel that follows):")
    for g, k in big:
        print(f"  {g / 1e3:7.1f}  {k[:90]}")


if __name__ == "__main__":
    main()
