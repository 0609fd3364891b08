"""Flags kernels in `hipcc -S --cuda-device-only` output whose scatter loops wait between stores.

    hipcc -O3 --offload-arch=gfx950 -x hip --cuda-device-only -S csrc/msd.hip -o msd.s && python tools/isa_store_wait_audit.py msd.s

A `s_waitcnt vmcnt(..)` that sits between two global stores with no load or atomic in between makes the second store wait
for the first (loads and stores share the counter): DESIGN.md 4.4, round 3.  Prints <file> <occurrences> <kernel>."""
import sys,re
for path in sys.argv[1:]:
    lines=open(path).read().split('\n')
    i=0
    cur=None
    res={}
    state=None  # after a store: 'S'; after S then wait: 'SW'
    for l in lines:
        s=l.strip()
        m=re.match(r'^(_Z\w+):',l)
        if m: cur=m.group(1); state=None; continue
        if not cur or not s or s.startswith(';'): continue
        op=s.split()[0]
        if op.startswith('global_store') or op.startswith('buffer_store'):
            if state=='SW': res[cur]=res.get(cur,0)+1
            state='S'
        elif op.startswith(('global_load','global_atomic','buffer_load','flat_load','flat_atomic','buffer_atomic')):
            state=None
        elif op=='s_waitcnt' and 'vmcnt(' in s:
            if state=='S': state='SW'
        elif op=='s_endpgm': state=None
    for k,v in res.items():
        if v>=2: print(path.split('/')[-1], v, k[:110])
