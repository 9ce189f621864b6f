"""Fold the per-kernel FETCH_SIZE / WRITE_SIZE summaries of scripts/gpu_pmc.sh into profiles/pmc_gemm_traffic.json
(the file bench.py reads for roofline.traffic): python scripts/pmc_summary.py <fetch.csv> <write.csv> <out.json>.
GEMM instantiations are grouped as bench.py names them (tile shape + operand layouts, the EDGE / k-group parameters
wild-carded); MSDA kernels keep their template arguments."""
import csv, json, re, sys


def key(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'(rscotr::gemm_f32_kernel<\d+, \d+, \d+, \d+, \w+, \w+), .*>', name)
    if m:
        return m.group(1) + ', *>'
    m = re.match(r'(rscotr::gemm_(?:bf16x6|h3)_kernel<\d+, \d+, \w+, \w+), .*>', name)
    if m:
        return m.group(1) + ', *>'
    m = re.match(r'rscotr::gemm_h3_128_kernel<(\w+, \w+), .*>', name)  # (bench.py names the 128 x 128 instantiation by its tile)
    return 'rscotr::gemm_h3_kernel<128, 128, ' + m.group(1) + ', *>' if m else name


def load(path):
    out = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            a = out.setdefault(key(r['Kernel_Name']), [0, 0.0])
            a[0] += int(r['Dispatches'])
            a[1] += float(r['Sum'])
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
kernels = {}
for k in fetch:
    if k in write:
        kernels[k] = dict(dispatches=fetch[k][0], fetch_kib_per_launch=fetch[k][1] / fetch[k][0],
                          write_kib_per_launch=write[k][1] / write[k][0])
json.dump({
    'command': "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --kernel-include-regex 'gemm_f32_kernel|gemm_h3|gemm_bf16x6|gemm_f32_group|msda_|attn_' -- python "
               'bench.py --steps 1 --warmup 0 --no-cpu-baseline --roofline-rounds 1 (scripts/gpu_pmc.sh, scripts/pmc_summary.py)',
    'units': 'KiB per dispatch as reported; FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM): '
             'bench.py doubles it',
    'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1]['dispatches'] * kv[1]['fetch_kib_per_launch'])),
}, open(sys.argv[3], 'w'), indent=1)
