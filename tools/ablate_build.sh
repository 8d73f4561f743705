#!/bin/bash
# One-translation-unit ablation build: recompile ONE .hip with extra -D flags and link it with the product's other objects.
#   tools/ablate_build.sh <tag> <file.hip> "<extra flags>"   ->  arrow-rs_amd/lib/ablate/libarrow_hip_<tag>.so   (load with AH_LIB_PATH)
set -e
cd "$(dirname "$0")/../arrow-rs_amd/csrc"
tag=$1; src=$2; extra=$3
make -j16 >/dev/null
mkdir -p build_ab ../lib/ablate
obj=build_ab/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-pass-failed $extra -c $src -o $obj
others=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libarrow_hip_$tag.so $obj $others
echo ../lib/ablate/libarrow_hip_$tag.so
