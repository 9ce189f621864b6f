#!/bin/bash
# builds scripts/lab/msda_lab (current csrc/msda.hip with the phase timers) and, given an older msda.hip as $1, msda_lab_old
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -I rscotr_amd/csrc -I include -w"
hipcc $F -c rscotr_amd/csrc/abi.hip -o /tmp/abi_lab.o || exit 1
hipcc $F -c scripts/lab/msda_lab.hip -o /tmp/msda_lab.o && hipcc --offload-arch=gfx950 /tmp/msda_lab.o /tmp/abi_lab.o -o scripts/lab/msda_lab || exit 1
if [ -n "$1" ]; then
  cp "$1" /tmp/msda_old.h
  hipcc $F -DMSDA_LAB_OLD -include /tmp/msda_old.h -c scripts/lab/msda_lab.hip -o /tmp/msda_lab_old.o && hipcc --offload-arch=gfx950 /tmp/msda_lab_old.o /tmp/abi_lab.o -o scripts/lab/msda_lab_old || exit 1
fi
