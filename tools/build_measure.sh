#!/bin/bash
# Builds the MEASUREMENT library build/measure/libbsvd_hip.so: the product sources + -DBSVD_MEASURE, i.e. with the kernel variants
# DESIGN.md 4.1d records as slower (F(4,3), 4-wave workgroups, forced tiles, the persistent form, conv3x3_wino.hip).  Never loaded by
# the product (bsvd_amd/_lib.py loads bsvd_amd/libbsvd_hip.so); tests marked `measure` and the record scripts select it with
# BSVD_HIP_LIB.  usage: tools/build_measure.sh ["<extra flags>"]  [suffix]   -> build/measure/libbsvd_hip<suffix>.so
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build/measure"
SUF="${2:-}"
EXTRA_HIPCC_FLAGS="-DBSVD_MEASURE $1" BSVD_OBJ_SUFFIX=_measure$SUF BSVD_OUT="$ROOT/build/measure/libbsvd_hip$SUF.so" "$ROOT/bsvd_amd/csrc/build.sh"
