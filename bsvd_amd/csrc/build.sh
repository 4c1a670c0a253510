#!/bin/bash
# Builds libbsvd_hip.so for gfx950 in-tree (bsvd_amd/libbsvd_hip.so).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
pids=()
for src in conv3x3_mfma conv3x3_edge_f32 bsvd_abi; do
  if [ ! -f "$HERE/obj/$src.o" ] || [ "$HERE/$src.hip" -nt "$HERE/obj/$src.o" ] || \
     [ "$HERE/bsvd_internal.h" -nt "$HERE/obj/$src.o" ] || [ "$ROOT/include/bsvd_hip.h" -nt "$HERE/obj/$src.o" ]; then
    $HIPCC $FLAGS ${EXTRA_HIPCC_FLAGS} -c "$HERE/$src.hip" -o "$HERE/obj/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$HERE"/obj/*.o -o "$ROOT/bsvd_amd/libbsvd_hip.so"
echo "built $ROOT/bsvd_amd/libbsvd_hip.so"
