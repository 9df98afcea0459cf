"""Prints a kernel's instruction-level event sequence from hipcc -S output: LD (global load), DMA (global_load_lds), ST, AT,
dr / dw (LDS read / write), M (MFMA run), BAR, W[v<vmcnt> l<lgkmcnt>] waits, br branches, basic-block labels.
Usage: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -I vmrframe_amd/csrc -I include X.hip -o X.s
       python scratch/isa_events.py X.s <substring of the mangled kernel name> [how many matches]
Used for DESIGN.md section 3.1c (compiler-inserted vmcnt(0) in front of transposed LDS reads)."""
import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]
names=[n for n in re.findall(r'^(\S+):\s*; @', s, re.M) if pat in n]
for name in names[:int(sys.argv[3]) if len(sys.argv)>3 else 1]:
    i=s.index(name+':'); j=s.find('s_endpgm', i); body=s[i:j].split('\n')
    out=[]
    def push(tag):
        if out and out[-1].split('x')[0]==tag:
            parts=out[-1].split('x'); n=int(parts[1]) if len(parts)>1 else 1; out[-1]='%sx%d'%(tag,n+1)
        else: out.append(tag)
    for ln in body:
        t=ln.strip()
        if re.match(r'(global_load|buffer_load)', t): push('LD' if 'lds' not in t else 'DMA')
        elif t.startswith('global_store') or t.startswith('buffer_store'): push('ST')
        elif t.startswith('global_atomic'): push('AT')
        elif t.startswith('ds_read') or t.startswith('ds_load'): push('dr')
        elif t.startswith('ds_write') or t.startswith('ds_store'): push('dw')
        elif 's_waitcnt' in t:
            m=re.search(r'vmcnt\((\d+)\)',t); l=re.search(r'lgkmcnt\((\d+)\)',t)
            out.append('W[%s%s]'%('v'+m.group(1) if m else '', ('l'+l.group(1)) if l else ''))
        elif t.startswith('s_barrier'): out.append('BAR')
        elif t.startswith('s_cbranch') or t.startswith('s_branch'): out.append('br')
        elif re.match(r'\.LBB\d+_\d+:',t): out.append('\n'+t.split(':')[0]+':')
        elif 'mfma' in t: push('M')
    print(name[:90]); print(' '.join(out))
