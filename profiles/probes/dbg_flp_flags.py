import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
from test_framelane_pair import _run_batch
fmt='Y8'; sw,sh,tw,th=160,90,219,123
ofmt, gfmt = O.FORMATS[fmt], pkg.FORMATS[fmt]
of = O.OracleFilter(ofmt, sw, sh, tw, th)
f = pkg.Filter(gfmt, sw, sh, tw, th, device=0)
srcs=[O.lcg_frame(ofmt, sw, sh, seed=100+k) for k in range(128)]
got=_run_batch(torch, pkg, f, gfmt, srcs, 128, 0)
print('kernel', f.last_kernel(0))
want=of.get_frame(srcs[0], threads=4)[0][:th,:tw]
g0=got[0][0][:th,:tw]
bad=np.argwhere(g0!=want)
print('bad', len(bad))
sx,sy,ids=f.plan_dump(0); ids=ids.reshape(th,tw); sets=f.plan_sets(0)
fs=f.plan_info(0).filter_size
seen=set()
for (y,x) in bad[:400]:
    s=ids[y,x]
    if s in seen: continue
    seen.add(s)
    m=sets[s].reshape(fs,fs)
    if len(seen)<=4:
        print('pixel',x,y,'set',s,'start',sx[x],sy[y]); print((m!=0).astype(int)); print(np.array2string(m,precision=4))
ys,xs=bad[:,0],bad[:,1]
print('x hist edges', np.bincount(np.minimum(xs,tw-1-xs))[:8], 'y hist edges', np.bincount(np.minimum(ys,th-1-ys))[:8])
# are all pixels of a bad set bad?
import os
print('JINC_FL_SKIP', os.environ.get('JINC_FL_SKIP'))
for fmt2 in ('Y16','Y32'):
    of2 = O.OracleFilter(O.FORMATS[fmt2], sw, sh, tw, th)
    f2 = pkg.Filter(pkg.FORMATS[fmt2], sw, sh, tw, th, device=0)
    srcs2=[O.lcg_frame(O.FORMATS[fmt2], sw, sh, seed=100+k) for k in range(128)]
    got2=_run_batch(torch, pkg, f2, pkg.FORMATS[fmt2], srcs2, 128, 0)
    want2=of2.get_frame(srcs2[0], threads=4)[0][:th,:tw]
    a=got2[0][0][:th,:tw]; 
    print(fmt2, f2.last_kernel(0), 'bad', int((a.view(np.uint32 if fmt2=='Y32' else a.dtype)!=want2.view(np.uint32 if fmt2=='Y32' else want2.dtype)).sum()))
