#!/bin/bash
# Build libm324.so of a git revision (default HEAD) into tools/lablibs/libm324_<name>.so for interleaved A/B runs:
#   tools/build_head_lib.sh [rev] [name]
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; name=${2:-head}
d=$(mktemp -d)
mkdir -p $d/motion324_amd/csrc $d/include tools/lablibs
for f in $(git ls-tree --name-only $rev motion324_amd/csrc/); do git show $rev:$f > $d/$f; done
git show $rev:include/m324.h > $d/include/m324.h
cd $d/motion324_amd/csrc
objs=()
for src in *.hip; do
    flags="-mllvm -amdgpu-mfma-vgpr-form"; [ "$src" = gemm_ring4.hip ] && flags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize $flags -c $src -o ${src%.hip}.o &
    objs+=(${src%.hip}.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OLDPWD/tools/lablibs/libm324_$name.so "${objs[@]}"
cd $OLDPWD; rm -rf $d
echo tools/lablibs/libm324_$name.so
