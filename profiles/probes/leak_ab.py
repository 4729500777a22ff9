"""Where does device memory go over create / frame / free cycles?  (round 6 diagnosis: tests/test_gpu_parity.py::test_create_free_cycles...)"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
kinds = [("Y8", 192, 108, 384, 216, {}), ("Y16", 160, 90, 219, 123, {}), ("Y8", 96, 64, 192, 128, dict(tap=12))]
def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
for mode in (None, 0, 1):
    for (fmt, sw, sh, tw, th, kw) in kinds:
        src = O.lcg_frame(O.FORMATS[fmt], sw, sh)
        rows = []
        for i in range(4):
            a = free()
            f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
            if mode is not None:
                f.set_pipeline(1, mode)
            b = free()
            f.get_frame(src)
            c = free()
            f.get_frame(src)
            d = free()
            f.close()
            g = free()
            rows.append((round(a - b, 1), round(b - c, 1), round(c - d, 1), round(g - d, 1), round(a - g, 1)))
        print("pin mode", mode, fmt, tw, th, kw, "MiB taken by [create, frame 1, frame 2], given back by close, lost per cycle:", rows, flush=True)
