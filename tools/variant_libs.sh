#!/bin/bash
# Product-flavour library variants that differ in ONE source file's -D knob (same-box A/B of a kernel constant):
#   bash tools/variant_libs.sh build rowchain RC_RD "8 12 16"     -> live2diff_amd/ablate/libl2d_<file>_<knob><value>.so   (CPU, here)
#   L2D_LIB=live2diff_amd/ablate/libl2d_rowchain_RC_RD16.so python ...                                                     (GPU box)
set -e
cd "$(dirname "$0")/.."
C=live2diff_amd/csrc; O=live2diff_amd/ablate; mkdir -p $O
[ "$1" = build ] || { echo "usage: $0 build <file> <KNOB> \"<values>\" [extra flags]"; exit 1; }
F=$2; K=$3; VALS=$4; EXTRA=$5
make -C $C -j8 2>&1 | tail -1
for v in $VALS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-unused-value -D$K=$v $EXTRA \
    -Xclang -target-feature -Xclang -packed-fp32-ops -c $C/$F.hip -o $O/${F}_$K$v.o 2>&1 | grep -v "not a recognized\|hip-link" &
done; wait
for v in $VALS; do
  objs=$(ls $C/*.o | grep -v probes | grep -v "/$F.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libl2d_${F}_$K$v.so $objs $O/${F}_$K$v.o
done
rm -f $O/*.o; ls -la $O
