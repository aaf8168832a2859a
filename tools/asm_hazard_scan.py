#!/usr/bin/env python
"""Build container: audit the LDS-DMA asm statements of wx_gemm8p.h in a -save-temps .s file.  hipcc treats an asm statement as one opaque
instruction and pads no hazards inside it (cdna_hip_programming.md 5.7 item 2): an SGPR written by a VALU instruction (v_readlane of a
spilled SGPR, v_readfirstlane) needs 5 wait states before a VMEM instruction reads it as base or descriptor.  The scan lists every
buffer_load / global_load_lds in gemm8p_kernel instantiations whose SGPR operands were VALU-written fewer than 5 states earlier.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/gemm8p_probe.hip -o /tmp/p -save-temps=obj
    python tools/asm_hazard_scan.py /tmp/gemm8p_probe-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

s = open(sys.argv[1]).read()
bad = tot = 0
for m in re.finditer(r'^(_ZN2wx13gemm8p_kernel\w+):', s, re.M):
    name = m.group(1)
    i = m.end()
    j = s.index('.Lfunc_end', i)
    lines = [l.strip() for l in s[i:j].split('\n') if l.startswith('\t') and not l.strip().startswith((';', '.'))]
    for n, l in enumerate(lines):
        if not l.startswith(('buffer_load_dwordx4', 'global_load_lds_dwordx4')):
            continue
        tot += 1
        used = set()
        for a, b in re.findall(r's\[(\d+):(\d+)\]', l):
            used |= set(range(int(a), int(b) + 1))
        for k in range(1, 7):
            if n - k < 0:
                break
            mm = re.match(r'(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)', lines[n - k])
            if mm and int(mm.group(2)) in used:
                ws = 0
                for q in lines[n - k + 1:n]:
                    mn = re.match(r's_nop (\d+)', q)
                    ws += (int(mn.group(1)) + 1) if mn else 1
                if ws < 5:
                    bad += 1
                    print(name[20:70], 'HAZARD', lines[n - k], '->', l, 'wait states', ws)
print(f'{tot} LDS-DMA instructions scanned, {bad} VALU-written-SGPR hazards')
sys.exit(1 if bad else 0)
