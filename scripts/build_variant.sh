#!/bin/bash
# A/B build of the library with extra compiler flags on ONE source: scripts/build_variant.sh <name> <source.hip> "<flags>"
#   -> rscotr_amd/_ab/lib_<name>.so (the other objects are the current rscotr_amd/_obj/*.o; run the normal build first);
#   use with RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_<name>.so (scripts/gpu_ab_bench.sh)
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p rscotr_amd/_ab
obj=rscotr_amd/_ab/${name}_${src}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-fast-math -Wno-unused-result $flags -x hip -c rscotr_amd/csrc/$src -o $obj -I rscotr_amd/csrc -I include || exit 1
objs=$(ls rscotr_amd/_obj/*.o | grep -v "/${src}.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o rscotr_amd/_ab/lib_${name}.so && echo rscotr_amd/_ab/lib_${name}.so
