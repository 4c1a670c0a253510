cd $GRAFT_REPO_ROOT
for i in 0 1; do echo "== ab$i $(sed -n "$((i+1))p" build/ab/variants.txt)"; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python tools/debug/wino_f32_bench.py 2 wino2 2>&1 | grep -v "amdgpu.ids\|BSVD_HIP_LIB"; done
