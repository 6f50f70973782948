#!/bin/bash
# register / scratch / occupancy figures of every kernel in the library (compile-time remarks)
cd "$(dirname "$0")/../raytracer_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -c rt_device.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        import subprocess
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0][-70:]; d={}
    for key in ('VGPRs:','AGPRs','ScratchSize','Occupancy','SGPRs Spill','VGPRs Spill','LDS Size'):
        m2=re.search(key+r'[^:]*:\s*(\d+)',l) if key!='VGPRs:' else re.search(r' VGPRs: (\d+)',l)
        if m2 and cur: d[key]=m2.group(1)
    if 'LDS Size' in l and cur:
        print(f'{cur:72s}', ' '.join(f'{k}={v}' for k,v in d.items())); cur=None
"
