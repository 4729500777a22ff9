import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
fmt='Y32'; sw,sh,tw,th=200,150,400,300
of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=31337)
want = of.get_frame(src, threads=4)[0][:th,:tw]
f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
for mode in (13, 2, 3, 15):
    f.set_kernel_mode(mode)
    got = f.get_frame(src)[0][:th,:tw]
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    print('mode', mode, f.last_kernel(0), 'bad', len(bad))
    if len(bad):
        ys, xs = bad[:,0], bad[:,1]
        print(' x%4 hist', np.bincount(xs % 4, minlength=4), 'y%2 hist', np.bincount(ys % 2, minlength=2), 'xrange', xs.min(), xs.max(), 'yrange', ys.min(), ys.max())
        y,x = bad[0]
        print(' first', x, y, got[y,x], want[y,x], 'neighbours want', want[y, x-2:x+3], 'got', got[y, x-2:x+3])
