#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <python args>   (PMC pass: kernel-trace + counters only)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$out" -o pmc --output-format csv -- python3 "$@"
