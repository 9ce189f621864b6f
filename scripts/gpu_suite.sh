#!/bin/bash
# Full -m gpu suite + default bench on the GPU box; logs under gpurun_out/$1 (default: suite).
tag=${1:-suite}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -150 > $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/pytest.log
cat $out/bench.json
