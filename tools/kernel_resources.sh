#!/bin/bash
# Developer tool: per-kernel register / spill / LDS figures of one source file, as hipcc reports them
#   bash tools/kernel_resources.sh rotate.hip [name filter]
cd "$(dirname "$0")/../ct_pvae_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr_$$.o 2>&1 |
  python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)};rows.append(cur)
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('vspill',r'VGPR Spill: (\d+)'),('sspill',r'SGPRs Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=int(m.group(1))
import subprocess
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip().split('(')[0]
    if flt in name: print('%-70s vgpr %3d sgpr %3d spill v%d s%d scratch %d occ %d lds %d'%(name[:70],r.get('vgpr',-1),r.get('sgpr',-1),r.get('vspill',0),r.get('sspill',0),r.get('scratch',0),r.get('occ',0),r.get('lds',0)))
" "$2"
rm -f /tmp/kr_$$.o
