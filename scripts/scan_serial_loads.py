#!/usr/bin/env python3
"""Finds kernels whose global loads go to memory ONE AT A TIME: a load issued with nothing in flight and waited for (s_waitcnt vmcnt(0))
before the next one is issued.  That is what a masked / selected / branched-on value right behind its load compiles to, and it halves the
bytes in flight per CU (round 4: the head / tail masks of the Planar register kernels at odd heights, 42 -> 60 % of the HBM peak once all
loads of a tile were issued before the first mask; the staging loops of the matrix pullbacks, 13 -> 23 %).

Usage (no GPU needed):
  mkdir -p /tmp/asm; cd bijectors.jl_amd/csrc
  for f in bjx_*.hip; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $(make -s -p 2>/dev/null | sed -n "s/^FLAGS_${f%.hip} := //p") \
      --offload-device-only -S $f -o /tmp/asm/${f%.hip}.s & done; wait
  python scripts/scan_serial_loads.py fam      # worst instantiation per kernel family;  no argument: every instantiation with >= 4 such loads
Columns: fraction of the kernel's loads that are serial, their count, all loads, the most loads ever in flight."""
import re,sys,collections,subprocess,glob
res=[]
for f in sorted(glob.glob('/tmp/asm/*.s')):
    cur=None
    st=None
    def flush():
        if cur and st and st['loads']>=6: res.append((st['serial'], st['loads'], st['maxout'], cur, f.split('/')[-1]))
    for line in open(f):
        m=re.match(r'^(_Z\S+):',line)
        if m:
            flush(); cur=m.group(1); st=dict(loads=0,out=0,serial=0,maxout=0,alone=False); continue
        if not cur: continue
        m=re.match(r'^\t([a-z_0-9]+)\s*(.*)',line)
        if not m: continue
        op,args=m.group(1),m.group(2)
        if op.startswith('global_load') or op.startswith('buffer_load'):
            st['alone'] = (st['out']==0)
            st['loads']+=1; st['out']+=1; st['maxout']=max(st['maxout'],st['out'])
        elif op=='s_waitcnt':
            mm=re.search(r'vmcnt\((\d+)\)',args)
            if mm:
                n=int(mm.group(1))
                if n==0 and st['out']==1 and st['alone']: st['serial']+=1
                st['out']=min(st['out'],n)
                st['alone']=False
    flush()
names=[r[3] for r in res]
dem=subprocess.run(['c++filt'],input="\n".join(names),capture_output=True,text=True).stdout.splitlines()
out=[]
for (serial,loads,maxout,nm,f),d in zip(res,dem):
    d=d.replace('(anonymous namespace)::','').replace('void ','')
    out.append((serial/loads, serial, loads, maxout, d.split('(')[0][:100], f))
mode=sys.argv[1] if len(sys.argv)>1 else 'all'
out.sort(reverse=True)
seen=set()
for frac,serial,loads,maxout,base,f in out:
    if serial<4: continue
    k=base.split('<')[0]
    if mode=='fam':
        if k in seen: continue
        seen.add(k)
    print(f"{frac:5.2f} serial={serial:3d} loads={loads:3d} maxinflight={maxout:3d}  {base}  [{f[4:-2]}]")
