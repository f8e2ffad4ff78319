#!/bin/bash
# Analysis builds of the pipelined flash loop with one component removed each (FAR_X masks in flash_attn_ring.hip; results are NOT
# attention): what does each component cost IN SITU?   bash tools/flash_ablate.sh build   (here, CPU)   /   ... run   (GPU box)
set -e
cd "$(dirname "$0")/.."
C=live2diff_amd/csrc; O=live2diff_amd/ablate; mkdir -p $O
MASKS="${MASKS:-0 1 2 4 8 16 3 7 15 31}"; KNOB="${KNOB:-FAR_X}"   # KNOB=FAR_V: schedule choices instead of ablations
if [ "$1" = build ]; then
  if [ -n "$PRODUCT" ]; then make -C $C -j8 2>&1 | tail -1; PF=""; SUF=".o"; else make -C $C PROBES=1 -j8 2>&1 | tail -1; PF="-DL2D_PROBES"; SUF=".probes.o"; fi
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value $PF -D$KNOB=$m \
      -Xclang -target-feature -Xclang -packed-fp32-ops -fno-honor-nans -c $C/flash_attn_ring.hip -o $O/far_x$m.o &
  done; wait
  for m in $MASKS; do
    objs=$(ls $C/*$SUF | grep -v flash_attn_ring | if [ -n "$PRODUCT" ]; then grep -v probes; else cat; fi)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libl2d_x$m.so $objs $O/far_x$m.o
  done
  rm -f $O/*.o; ls -la $O
else
  for m in $MASKS; do
    echo "== $KNOB=$m"; L2D_LIB=$O/libl2d_x$m.so python tools/flash_time.py 4 2>&1 | grep -v amdgpu.ids | head -${ROWS:-1}
  done
fi
