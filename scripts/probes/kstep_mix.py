"""Instruction mix of a register-resident kernel per k step: usage  python scripts/probes/kstep_mix.py <hipcc -save-temps gfx950 .s file> <kernel name regex>
Prints, for the k steps between consecutive s_barrier instructions, how many instructions stand behind each MFMA -- what showed that the
epilogue of csrc/rchain.hip was being sunk into the last gaps of a step."""
import re,sys
from collections import Counter
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
starts=[i for i,l in enumerate(lines) if re.match(r'^_ZN.*'+pat+'.*:', l)]
ends=[i for i,l in enumerate(lines) if l.strip().startswith('s_endpgm')]
def is_instr(l):
    l=l.strip()
    return l and not l.startswith(('.',';','//')) and not re.match(r'^[\.\w$]+:', l)
for st in starts:
    en=min(e for e in ends if e>st)
    body=[l.strip() for l in lines[st+1:en] if is_instr(l)]
    bar=[i for i,l in enumerate(body) if l.startswith('s_barrier')]
    segs=[(bar[i],bar[i+1]) for i in range(len(bar)-1)]
    print(lines[st][30:62], 'total', len(body), [b-a for a,b in segs[4:10]], 'scratch ops', sum(1 for l in body if 'scratch_' in l))
    for a,b in segs[6:8]:
        runs=[];cur=0
        for l in body[a:b]:
            if l.startswith('v_mfma'): runs.append(cur); cur=0
            else: cur+=1
        c=Counter(x.split()[0] for x in body[a:b])
        print('   between MFMAs:', runs, 'waitcnt', c.get('s_waitcnt',0), 'nop', c.get('s_nop',0))
