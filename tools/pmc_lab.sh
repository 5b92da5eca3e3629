#!/bin/bash
# usage: tools/pmc_lab.sh <outdir> <counters...> -- <gemm_lab args>   (PMC pass over the C++ GEMM lab)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$out" -o pmc --output-format csv -- tools/gemm_lab "$@" > "$out.log" 2>&1
