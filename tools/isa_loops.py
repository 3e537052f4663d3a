#!/usr/bin/env python3
"""Static loop structure of one kernel in an llvm-objdump listing (tools/kernel_resources.sh with KEEP=file):
every backward branch is a loop; prints its length and instruction mix.  usage: isa_loops.py listing.s kernel-substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
ksub = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l) and ksub in l][0]
end = [i for i, l in enumerate(lines) if i > start and re.match(r'^[0-9a-f]+ <', l)]
end = end[0] if end else len(lines)
ins = []
for l in lines[start + 1:end]:
    m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):', l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
idx = {a: i for i, (a, _, _) in enumerate(ins)}
def mix(seg):
    nv = sum(1 for x in seg if x[1].startswith('v_'))
    f64 = sum(1 for x in seg if x[1].startswith('v_') and ('f64' in x[1] or 'u64' in x[1] or 'b64' in x[1]))
    q = sum(1 for x in seg if re.match(r'v_(rcp|rsq|sqrt|mul_lo_u32|mul_hi_u32|mad_u64_u32|div_scale|div_fmas|div_fixup)', x[1]))
    ns = sum(1 for x in seg if x[1].startswith('s_') and not x[1].startswith('s_load'))
    nm = sum(1 for x in seg if x[1].startswith(('global_', 'scratch_', 'flat_', 'buffer_')))
    sc = sum(1 for x in seg if x[1].startswith('scratch_'))
    nsm = sum(1 for x in seg if x[1].startswith('s_load'))
    lds = sum(1 for x in seg if x[1].startswith('ds_'))
    return "valu %4d (64-bit %3d, slow %3d) salu %4d vmem %3d (scratch %2d) smem %3d lds %3d" % (nv, f64, q, ns, nm, sc, nsm, lds)
print("kernel: %d instructions; %s" % (len(ins), mix(ins)))
loops = []
for i, (a, op, args) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        off = int(args.split()[0])
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt in idx and idx[tgt] <= i:
            loops.append((idx[tgt], i))
for s, e in sorted(loops):
    print("loop %5d..%5d (%5d instr)  %s" % (s, e, e - s + 1, mix(ins[s:e + 1])))
