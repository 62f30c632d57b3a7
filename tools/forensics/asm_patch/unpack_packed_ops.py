import re, sys
# usage: unpack.py out.s MODE   MODE: all | global | lds | k0 | k1 | first:N (first N pk ops) | range:a:b
src=open('dev.s').read().split('\n')
start=next(i for i,l in enumerate(src) if l.startswith('_Z17warp_tiled_kernelILi3EDF16bE') and l.rstrip().endswith('PKfS1_PfiiiPvii'))
end=next(i for i in range(start,len(src)) if src[i].startswith('.Lfunc_end'))
pk=re.compile(r'^\s+v_pk_(mul|add)_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)$')
def sel(s, name, default):
    m=re.search(name+r':\[(\d),(\d)\]', s)
    return (int(m.group(1)),int(m.group(2))) if m else default
mode=sys.argv[2]
idx=-1; out=src[:start]; n=0
lo_hi=None
if mode.startswith('range:') or mode.startswith('keep:'): lo_hi=tuple(int(x) for x in mode.split(':')[1:])
for l in src[start:end]:
    m=pk.match(l)
    if m:
        idx+=1
        take = mode=='all' or (mode.startswith('range:') and lo_hi[0]<=idx<lo_hi[1]) or (mode.startswith('keep:') and not (lo_hi[0]<=idx<lo_hi[1]))
        if take:
            op,d0,d1,a0,a1,b0,b1,rest=m.groups()
            os_=sel(rest,'op_sel',(0,0)); oh=sel(rest,'op_sel_hi',(1,1))
            A=(a0,a1); B=(b0,b1)
            out.append(f'\tv_{op}_f32_e32 v120, v{A[os_[0]]}, v{B[os_[1]]}')
            out.append(f'\tv_{op}_f32_e32 v121, v{A[oh[0]]}, v{B[oh[1]]}')
            out.append(f'\tv_mov_b32_e32 v{d0}, v120')
            out.append(f'\tv_mov_b32_e32 v{d1}, v121')
            n+=1
            continue
    out.append(l)
out+=src[end:]
open(sys.argv[1],'w').write('\n'.join(out)); print(sys.argv[1], 'replaced', n, 'of', idx+1)
