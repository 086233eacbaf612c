#!/bin/bash
# per-kernel register / LDS usage of the gfx950 build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Iinclude -Iminbpe_amd/csrc minbpe_amd/csrc/bpe_api.hip -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    for k in ('VGPRs','SGPRs','ScratchSize','Occupancy','LDS Size'):
        m=re.search(k+r'[^:]*: (\d+)',line)
        if m and cur: rows[cur][k]=m.group(1)
for k,v in rows.items():
    print(k[:60].ljust(60), ' '.join(f'{a}={b}' for a,b in v.items()))
"
