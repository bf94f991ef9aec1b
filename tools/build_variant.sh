#!/bin/bash
# Builds a variant of the library with extra flags for the one-wave-per-example kernels:
#   [KIND=wave64|wave1d|lines|large] [NPS="4 8"] tools/build_variant.sh NAME "-DFOO -DBAR"   ->  wavenet_amd/lib/exp/NAME.so
# (load it with WAVENET_HIP_LIBRARY; tools/ab_bench.sh runs the bench with every library in that directory)
set -e
KIND=${KIND:-wave64}
NPS=${NPS:-"4 8"}
cd "$(dirname "$0")/../wavenet_amd/csrc"
mkdir -p ../lib/exp build/exp
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
SKIP=""
NEW=""
if [ "$KIND" = large ]; then
  # the streamed path (ranking and level kernels) is one translation unit without a tap count
  /opt/rocm/bin/hipcc $FLAGS $2 -c wn_large.hip -o build/exp/wn_large_$1.o
  SKIP="\|wn_large.o"
  NEW="build/exp/wn_large_$1.o"
  NPS=""
fi
for np in $NPS; do
  /opt/rocm/bin/hipcc $FLAGS $2 -DWN_NP=$np -c wn_${KIND}_np.hip -o build/exp/wn_${KIND}_np${np}_$1.o &
  SKIP="$SKIP\|wn_${KIND}_np${np}.o"
  NEW="$NEW build/exp/wn_${KIND}_np${np}_$1.o"
done
wait
OBJS=$(ls build/*.o | grep -v "nothing$SKIP")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/$1.so $OBJS $NEW -ldl
echo built ../lib/exp/$1.so
