#!/bin/bash
# per-kernel VGPR / SGPR / occupancy / scratch of the gfx950 build
/opt/rocm/bin/hipcc $RU_EXTRA -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -c raytracer_amd/csrc/rt_lib.hip -o /tmp/rt_lib_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur={}
rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur)
    for k in ('TotalSGPRs','VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(r' '+k+r': (\d+)',l)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'rtd::Cfg<([^>]*)>',lambda m:'Cfg<'+m.group(1).replace('u','')+'>',n)
    n=n.split('(')[0]
    print('%-70s sgpr %3s vgpr %3s scratch %3s occ %s lds %s'%(n[-70:],r.get('TotalSGPRs'),r.get('VGPRs'),r.get('ScratchSize \[bytes/lane\]'),r.get('Occupancy \[waves/SIMD\]'),r.get('LDS Size \[bytes/block\]')))
"
