#!/bin/bash
# A/B build of the WHOLE library with extra compiler flags: scripts/build_variant_all.sh <name> "<flags>" -> rscotr_amd/_ab/lib_<name>.so
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p rscotr_amd/_ab/$name
objs=""
for src in rscotr_amd/csrc/*.hip rscotr_amd/csrc/*.cpp; do
  o=rscotr_amd/_ab/$name/$(basename $src).o
  x="-x hip"; [[ $src == *.cpp ]] && x=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-fast-math -Wno-unused-result -w $flags $x -c $src -o $o -I rscotr_amd/csrc -I include &
  objs="$objs $o"
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o rscotr_amd/_ab/lib_${name}.so && echo rscotr_amd/_ab/lib_${name}.so
