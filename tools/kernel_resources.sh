#!/bin/bash
# Register / spill / LDS / occupancy table of every kernel of one build (compile-time: no GPU needed).
# usage: [EXTRA_HIPCC_FLAGS=...] tools/kernel_resources.sh   -> builds into a scratch object dir, prints one line per kernel
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$(mktemp -d)
SRCS="conv3x3_mfma conv3x3_winox conv3x3_edge_f32 bsvd_abi"
case " ${EXTRA_HIPCC_FLAGS} " in *" -DBSVD_MEASURE"*) SRCS="$SRCS conv3x3_wino";; esac      # measurement builds only (bsvd_amd/csrc/build.sh)
for src in $SRCS; do
  XF=""; [ "$src" = conv3x3_winox ] && XF="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/bsvd_amd/csrc -Wno-unused-function \
     $XF ${EXTRA_HIPCC_FLAGS} -Rpass-analysis=kernel-resource-usage -c $ROOT/bsvd_amd/csrc/$src.hip -o $OUT/$src.o 2> $OUT/$src.log &
done
wait
cat $OUT/*.log | python3 -c "
import re,sys,subprocess
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r'remark: [^ ]+ +(Function Name|Name): (\S+)',l)
    if m: cur={'name':m.group(2)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=n.replace('bsvd::','').replace('void ','').replace('(ConvParams)','')
    print('%-84s VGPR %3d AGPR %3d scratch %4d occ %d LDS %6d'%(n[:84],r.get('VGPRs',-1),r.get('AGPRs',-1),r.get('ScratchSize',-1),r.get('Occupancy',-1),r.get('LDS Size',-1)))
"
rm -rf $OUT
