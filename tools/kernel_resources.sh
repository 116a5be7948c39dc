#!/bin/bash
# Register / scratch / LDS usage of every traversal kernel of one .hip source (development aid; cross-compiles, no GPU).
# usage: tools/kernel_resources.sh csrc/trace_subdiv.hip [extra hipcc flags]
cd "$(dirname "$0")/../embree-compressed_amd" || exit 1
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -I../include --offload-arch=gfx950 --offload-device-only -c \
  -Rpass-analysis=kernel-resource-usage "$@" "$src" -o /tmp/kres.o 2> /tmp/kres.log
python3 - <<'PY'
import re, subprocess
cur = None
for line in open('/tmp/kres.log'):
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace('rtamd::dev::', '').replace('(rtamd::LaunchParams)', '').replace('void ', '')
        vals = {}
    for key in ('VGPRs', 'ScratchSize [bytes/lane]', 'LDS Size [bytes/block]', 'Occupancy [waves/SIMD]'):
        m = re.search(re.escape(key) + r': (\d+)', line)
        if m and cur:
            vals[key] = m.group(1)
            if key == 'LDS Size [bytes/block]':
                print('%-70s vgpr %3s scratch %4s occ %s lds %s' % (cur, vals.get('VGPRs'), vals.get('ScratchSize [bytes/lane]'), vals.get('Occupancy [waves/SIMD]'), vals[key]))
PY
