"""Per-basic-block instruction counts of a kernel in a hipcc -S listing (/tmp/w4.s): v_mfma / scratch_load / scratch_store /
v_accvgpr_* / buffer_store, and the instruction mix of the block with the most MFMAs (the K loop).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only cplxmodule_amd/csrc/gemm_bf16_w4.hip -o /tmp/w4.s
    python scripts/r05/isa_blocks.py w4p_kernelILb1ELb0ELb0 [more mangled-name fragments]"""
import re, collections, sys
s=open('/tmp/w4.s').read()
funcs=re.split(r'\n(?=_ZN7cplxamd2w4\w+:)', s)
for tag in sys.argv[1:]:
    for f in funcs:
        name=f.split(':',1)[0]
        if tag not in name: continue
        lines=f.split('\n')
        cur=None; stats=[]; ops={}
        for l in lines:
            m=re.match(r'^(\.LBB\d+_\d+):',l)
            if m: cur=[m.group(1),0,0,0,0,0,0]; stats.append(cur); ops[cur[0]]=[]; continue
            if cur is None: continue
            t=l.strip()
            if not t or t.startswith(';') or t.startswith('.'): continue
            cur[1]+=1
            op=t.split()[0]; ops[cur[0]].append(op)
            if op.startswith('v_mfma'): cur[2]+=1
            if op.startswith('scratch_load'): cur[3]+=1
            if op.startswith('scratch_store'): cur[4]+=1
            if op.startswith('v_accvgpr'): cur[5]+=1
            if op.startswith('buffer_store'): cur[6]+=1
        print(name[-48:]); print("label n mfma sld sst accv bst")
        for c in stats:
            if c[1]>20: print(c)
        if not stats:
            continue
        best=max(stats,key=lambda c:c[2])
        print(best[0], dict(collections.Counter(ops[best[0]]).most_common(16)))
