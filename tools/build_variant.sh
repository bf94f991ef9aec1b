#!/bin/bash
# Builds a variant of the library with extra flags for the one-wave-per-example kernels of 4 and 8 taps:
#   tools/build_variant.sh NAME "-DFOO -DBAR"   ->  wavenet_amd/lib/exp/NAME.so
# (load it with WAVENET_HIP_LIBRARY; tools/ab_bench.sh runs the bench with every library in that directory)
set -e
cd "$(dirname "$0")/../wavenet_amd/csrc"
mkdir -p ../lib/exp build/exp
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
for np in 4 8; do
  /opt/rocm/bin/hipcc $FLAGS $2 -DWN_NP=$np -c wn_wave64_np.hip -o build/exp/wn_wave64_np${np}_$1.o &
done
wait
OBJS=$(ls build/*.o | grep -v "wn_wave64_np4.o\|wn_wave64_np8.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/$1.so $OBJS build/exp/wn_wave64_np4_$1.o build/exp/wn_wave64_np8_$1.o -ldl
echo built ../lib/exp/$1.so
