#!/bin/bash
# Timing ablations of the fused FFN kernel (csrc/ffn.hip, FFN_ABL_* macros): build here (CPU), run on the GPU box.
#   bash scripts/lab/ffn_abl.sh build        -> rscotr_amd/_ab/lib_ffn_<variant>.so
#   bash scripts/lab/ffn_abl.sh run <tag>    -> gpurun_out/<tag>/ffn_abl.txt
cd "$(dirname "$0")/../.."
variants="sgb3:-DFFN_SGB=3 sgb2:-DFFN_SGB=2 nohid:-DFFN_ABL_NOHID"
if [ "$1" = build ]; then
  for v in $variants; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
    bash scripts/build_variant.sh ffn_$name ffn.hip "$flags" || exit 1
  done
else
  tag=${2:-ffn_abl}; mkdir -p gpurun_out/$tag
  out=gpurun_out/$tag/ffn_abl.txt; : > $out
  echo "base $(python scripts/lab/ffn_cold.py fused 2>/dev/null | tail -1)" >> $out
  for v in $variants; do
    name=${v%%:*}
    echo "$name $(RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_ffn_$name.so python scripts/lab/ffn_cold.py fused 2>/dev/null | tail -1)" >> $out
  done
  cat $out
fi
