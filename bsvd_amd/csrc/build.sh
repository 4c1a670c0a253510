#!/bin/bash
# Builds libbsvd_hip.so for gfx950 in-tree (bsvd_amd/libbsvd_hip.so).  hipcc cross-compiles without a GPU.
# Objects are rebuilt when a source / header is newer OR when the compile flags (EXTRA_HIPCC_FLAGS tuning defines
# included) differ from the ones the object was built with (recorded next to it in <obj>.flags).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function ${EXTRA_HIPCC_FLAGS}"
OUT="${BSVD_OUT:-$ROOT/bsvd_amd/libbsvd_hip.so}"
# objects live outside the package: build/obj (product), build/obj_ab<i> (tools/build_ab.sh variants)
OBJ="$ROOT/build/obj${BSVD_OBJ_SUFFIX:-}"
mkdir -p "$OBJ"
# the product library: four translation units.  A measurement build (EXTRA_HIPCC_FLAGS contains -DBSVD_MEASURE, tools/build_measure.sh)
# adds conv3x3_wino.hip (the rejected all-positions-per-wave Winograd kernel) and the variant instantiations of conv3x3_winox.hip
SRCS="conv3x3_mfma conv3x3_winox conv3x3_edge_f32 bsvd_abi"
case " ${EXTRA_HIPCC_FLAGS} " in *" -DBSVD_MEASURE"*) SRCS="$SRCS conv3x3_wino";; esac
pids=()
for src in $SRCS; do
  if [ ! -f "$OBJ/$src.o" ] || [ "$HERE/$src.hip" -nt "$OBJ/$src.o" ] || \
     [ "$HERE/bsvd_internal.h" -nt "$OBJ/$src.o" ] || [ "$HERE/wino_forms.h" -nt "$OBJ/$src.o" ] || [ "$ROOT/include/bsvd_hip.h" -nt "$OBJ/$src.o" ] || \
     [ "$(cat "$OBJ/$src.flags" 2>/dev/null)" != "$FLAGS" ]; then
    # conv3x3_winox: no SLP vectorizer (it turns the transform's fma_mix forms into convert + packed-fp32 math, see dec_pair)
    XF=""; [ "$src" = conv3x3_winox ] && XF="-fno-slp-vectorize"
    ( $HIPCC $FLAGS $XF -c "$HERE/$src.hip" -o "$OBJ/$src.o" && echo "$FLAGS" > "$OBJ/$src.flags" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for src in $SRCS; do OBJS="$OBJS $OBJ/$src.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
