#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export LONGK=1
echo "== model choice"; python scripts/bench_wplanes.py 2>&1 | grep -v amdgpu.ids
for f in 256,2 256,3 256,4 256,6 256,8 128,3 128,4 128,6 128,8; do echo "== force $f"; RSCOTR_WPLANES_FORCE=$f python scripts/bench_wplanes.py 2>&1 | grep -v amdgpu.ids | sed 's/hot: in-kernel.*planes/planes hot/; s/cold: in-kernel *[0-9.]* us/cold/'; done
