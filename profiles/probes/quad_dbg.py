import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
import __graft_entry__ as e
pkg=e.load_package(); O=e.load_oracle()
fmt="Y32"; sw,sh,tw,th=320,180,640,360
of=O.OracleFilter(O.FORMATS[fmt],sw,sh,tw,th,tap=3)
f=pkg.Filter(pkg.FORMATS[fmt],sw,sh,tw,th,device=0,tap=3)
src=O.lcg_frame(O.FORMATS[fmt],sw,sh,seed=6100)
want=of.get_frame(src,threads=4)[0][:th,:tw]
f.set_kernel_mode(13)
got=f.get_frame(src)[0][:th,:tw]
bad=(got.view(np.uint32)!=want.view(np.uint32))
print("bad total", bad.sum())
for py in (0,1):
    for px in (0,1):
        print("phase x%2=",px,"y%2=",py, bad[py::2,px::2].sum(), "of", bad[py::2,px::2].size)
ys,xs=np.nonzero(bad)
print("x range",xs.min(),xs.max(),"y range",ys.min(),ys.max())
y,x=ys[0],xs[0]
print("first",x,y,got[y,x],want[y,x], "neighbors want", want[y,x-1:x+2], "got", got[y,x-1:x+2])
# is got[y,x] equal to some want nearby?
for dy in range(-2,3):
    for dx in range(-2,3):
        if got[y,x]==want[y+dy,x+dx]: print("matches want at",dx,dy)
d=np.abs(got-want)[bad]; print("abs diff median",np.median(d),"max",d.max())
