"""The loop of tests/test_gpu_parity.py::test_create_free_cycles_do_not_leak_device_memory, per plan kind (round 6 diagnosis)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
kinds = [("Y8", 192, 108, 384, 216, {}), ("Y8", 192, 108, 288, 162, {}), ("YUV420P8", 256, 144, 128, 72, {}), ("Y16", 160, 90, 219, 123, {}), ("Y8", 96, 64, 192, 128, dict(tap=12))]
srcs = [O.lcg_frame(O.FORMATS[k[0]], k[1], k[2]) for k in kinds]
def cycle(n, which=None):
    for i in range(n):
        j = which if which is not None else i % len(kinds)
        fmt, sw, sh, tw, th, kw = kinds[j]
        f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
        f.get_frame(srcs[j])
        f.close()
cycle(10); torch.cuda.synchronize(); f0,_ = torch.cuda.mem_get_info()
cycle(60); torch.cuda.synchronize(); f1,_ = torch.cuda.mem_get_info()
print("all kinds: MiB lost over 60 cycles", (f0-f1)/2**20, flush=True)
for j in range(5):
    cycle(5, j); torch.cuda.synchronize(); f0,_ = torch.cuda.mem_get_info()
    cycle(30, j); torch.cuda.synchronize(); f1,_ = torch.cuda.mem_get_info()
    print("kind", j, "MiB lost over 30 cycles", (f0-f1)/2**20, flush=True)
