#!/bin/bash
# register / scratch use of the kernels of one csrc file whose mangled name matches $2:  scripts/lab/kernel_regs.sh msda.hip 'tile_kernelILi32ELi4'
cd "$(dirname "$0")/../.."
SRC=${3:-rscotr_amd/csrc/$1}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rscotr_amd/csrc -I include -w --cuda-device-only -S -x hip $SRC -o /tmp/kr.s || exit 1
python3 - "$2" <<'PY'
import re, sys
txt = open('/tmp/kr.s').read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', txt, re.S):
    blk = m.group(0)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if not re.search(sys.argv[1], name):
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print(f"{name[:90]}: vgpr {g('vgpr_count')} agpr {g('agpr_count')} sgpr {g('sgpr_count')} spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
PY
