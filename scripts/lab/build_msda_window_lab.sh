#!/bin/bash
# builds scripts/lab/msda_window_lab against the tree's librscotr.so (python -c 'import __graft_entry__ as g; g.build()' first)
cd "$(dirname "$0")/../.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -w scripts/lab/msda_window_lab.hip -L rscotr_amd -lrscotr -Wl,-rpath,'$ORIGIN/../../rscotr_amd' -o scripts/lab/msda_window_lab
