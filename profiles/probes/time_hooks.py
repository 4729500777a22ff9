#!/usr/bin/env python3
"""How long do the measurement hooks take (clock sampler start / stop, instruction-pair probe)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import __graft_entry__ as e
pkg = e.load_package()
torch.cuda.set_device(0)
t = time.perf_counter(); c = pkg.ClockSampler(0, 60.0); t1 = time.perf_counter(); time.sleep(0.2); g = c.stop(); t2 = time.perf_counter()
print("sampler start %.3f s, stop %.3f s, clocks %s" % (t1 - t, t2 - t1 - 0.2, g), flush=True)
for w in (4, 6, 8):
    t = time.perf_counter(); r = pkg.valu_pair_probe(0, w); print("pair probe w=%d: %.3f s -> %s" % (w, time.perf_counter() - t, r), flush=True)
