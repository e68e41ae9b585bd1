#!/bin/bash
# A/B library builds for same-box comparisons (scripts/dbg/ab_bench.sh): build_ab.sh <name> [-Dflags...]
# env BASE_REV=<git rev>: take the kernel sources of that revision instead of the working tree
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
src=tinyrecurrentunet_amd/csrc
inc=include
if [ -n "$BASE_REV" ]; then
  tmp=$(mktemp -d)
  mkdir -p $tmp/csrc $tmp/include
  for f in $(git ls-tree --name-only $BASE_REV $src/ | grep -E '\.(hip|hpp)$'); do git show $BASE_REV:$f > $tmp/csrc/$(basename $f); done
  git show $BASE_REV:include/trunet_hip.h > $tmp/include/trunet_hip.h
  src=$tmp/csrc; inc=$tmp/include
fi
mkdir -p scripts/dbg/ab
objs=""
for f in $src/*.hip; do
  o=/tmp/ab_${name}_$(basename $f .hip).o
  hipcc --offload-arch=gfx950 -O3 -fPIC -I$inc -I$src "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dbg/ab/lib_$name.so $objs
echo built scripts/dbg/ab/lib_$name.so
