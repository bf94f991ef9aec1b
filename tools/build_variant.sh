#!/bin/bash
# Builds a variant of the library with extra flags for wn_wave64.hip: tools/build_variant.sh NAME "-DFOO -DBAR"
set -e
cd "$(dirname "$0")/../wavenet_amd/csrc"
mkdir -p ../lib/exp build/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result $2 -c wn_wave64.hip -o build/exp/wn_wave64_$1.o
OBJS=$(ls build/*.o | grep -v wn_wave64)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/$1.so $OBJS build/exp/wn_wave64_$1.o -ldl
echo built ../lib/exp/$1.so
