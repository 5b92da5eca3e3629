#!/usr/bin/env python3
"""Kernel sequence (name, duration, gap to the previous kernel) of the LAST n dispatches in a rocprofv3 rocpd database.
usage: tools/rocpd_seq.py results.db [n]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = con.execute(f"select {name}, start, end from kernels order by start").fetchall()[-n:]
prev = None
for nm, s, e in rows:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {nm[:110]}")
    prev = e
