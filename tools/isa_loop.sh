#!/bin/bash
# dump the ISA of k_raster<DR=0,OBJ=0> and print the instruction mix of its env loop
cd /tmp && mkdir -p t && cd t
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I/root/repo/include -I/root/repo/gym-duckietown_amd/csrc -S --cuda-device-only -o render.s ${1:-/root/repo/gym-duckietown_amd/csrc/render.hip} 2>/dev/null
L=$(grep -n "^_ZN12_GLOBAL__N_18k_rasterILb0ELb0E.*:" render.s | cut -d: -f1)
awk -v s=$L "NR>=s" render.s | awk "/^.Lfunc_end/{exit} {print}" > kr00.s
python3 - <<'PY'
import re,collections
lines=open('/tmp/t/kr00.s').read().split('\n')
# loop = from the last "Inner Loop Header" label to the last line mentioning "in Loop"
hdr=[i for i,l in enumerate(lines) if 'Loop Header' in l][-1]
end=max(i for i,l in enumerate(lines) if 'in Loop: Header' in l)
# extend end to next label
j=end+1
while j<len(lines) and not lines[j].startswith('.LBB'): j+=1
body=[l.strip() for l in lines[hdr:j] if l.strip() and not l.strip().startswith((';','.'))]
ops=collections.Counter(l.split()[0] for l in body)
v=sum(c for o,c in ops.items() if o.startswith('v_'))
print("loop lines",len(body),"VALU",v,"SALU",sum(c for o,c in ops.items() if o.startswith('s_')),"VMEM",sum(c for o,c in ops.items() if o.startswith('global_')),"DS",sum(c for o,c in ops.items() if o.startswith('ds_')))
print(sorted(ops.items(), key=lambda kv:-kv[1])[:45])
PY
grep -E "NumVgprs|Occupancy|ScratchSize" render.s | sed -n '16,18p'
