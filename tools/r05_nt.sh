#!/bin/bash
out=gpurun_out/r5hp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
setsid timeout -s KILL 500 python3 tools/clip_ab.py M324_NT_MB=128,48,16 --rounds 5 > $out/clip_nt.txt 2>&1; tail -4 $out/clip_nt.txt
