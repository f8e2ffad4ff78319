#!/bin/bash
# Analysis builds of the pipelined flash loop with one component removed each (FAR_X masks in flash_attn_ring.hip; results are NOT
# attention): what does each component cost IN SITU?   bash tools/flash_ablate.sh build   (here, CPU)   /   ... run   (GPU box)
set -e
cd "$(dirname "$0")/.."
C=live2diff_amd/csrc; O=live2diff_amd/ablate; mkdir -p $O
MASKS="${MASKS:-0 1 2 4 8 16 32 64 3 7 15 31 96}"
if [ "$1" = build ]; then
  make -C $C PROBES=1 -j8 2>&1 | tail -2
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -DL2D_PROBES -DFAR_X=$m \
      -Xclang -target-feature -Xclang -packed-fp32-ops -fno-honor-nans -c $C/flash_attn_ring.hip -o $O/far_x$m.o &
  done; wait
  for m in $MASKS; do
    objs=$(ls $C/*.probes.o | grep -v flash_attn_ring)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libl2d_x$m.so $objs $O/far_x$m.o
  done
  rm -f $O/*.o; ls -la $O
else
  for m in $MASKS; do
    echo "== FAR_X=$m"; L2D_LIB=$O/libl2d_x$m.so python tools/flash_time.py 4 2>&1 | grep -v amdgpu.ids | head -${ROWS:-1}
  done
fi
