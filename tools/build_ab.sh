#!/bin/bash
# Builds variant libraries build/ab/lib_ab<i>.so in the container (one per flag string) for tools/ab_prebuilt.sh.
# usage: tools/build_ab.sh "<flags 0>" "<flags 1>" ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $ROOT/build/ab
rm -f $ROOT/build/ab/lib_ab*.so
i=0
for f in "$@"; do
  ( EXTRA_HIPCC_FLAGS="$f" BSVD_OBJ_SUFFIX=_ab$i BSVD_OUT=$ROOT/build/ab/lib_ab$i.so $ROOT/bsvd_amd/csrc/build.sh > /tmp/build_ab$i.log 2>&1 || echo "build $i failed: $f" ) &
  i=$((i+1))
done
wait
echo "$@" | tr ' ' '\n' > /dev/null
i=0; for f in "$@"; do echo "ab$i: $f" ; i=$((i+1)); done | tee $ROOT/build/ab/variants.txt
ls -la $ROOT/build/ab/
