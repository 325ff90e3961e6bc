#!/bin/bash
# upflow_pytorch_amd/libupflow_hip_alt.so = the current build with the given sources taken from a git ref (default HEAD): the
# "before" side of tools/ab_lib.sh / tools/ab_lib_train.sh.     bash tools/build_alt.sh HEAD sgu_blend [conv_wgrad ...]
set -e
REF=${1:-HEAD}; shift
R=$(pwd); P=$R/upflow_pytorch_amd; T=$(mktemp -d)
OBJS=""
for f in api corr81_fwd corr81_bwd conv3x3 conv_c8 conv_x3 conv_wgrad warp sgu_blend misc loss; do
  if [[ " $* " == *" $f "* ]]; then
    git show $REF:upflow_pytorch_amd/csrc/$f.hip > $T/$f.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -I$P/csrc -I$R/include -Xclang -target-feature -Xclang -packed-fp32-ops $( [[ " sgu_blend loss misc warp corr81_fwd corr81_bwd " == *" $f "* ]] && python -c "from upflow_pytorch_amd import _build; print(\" \".join(dict(_build.SOURCES)[\"$f.hip\"]))" ) -c $T/$f.hip -o $T/$f.o 2>/dev/null
    OBJS="$OBJS $T/$f.o"
  else
    OBJS="$OBJS $P/build/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libupflow_hip_alt.so $OBJS
rm -rf $T; ls -la $P/libupflow_hip_alt.so
