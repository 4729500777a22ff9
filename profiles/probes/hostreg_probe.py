#!/usr/bin/env python3
"""Cost of hipHostRegister / hipHostUnregister on the GPU box's host, by buffer size and with several threads registering
at once (each its own buffers).  Decides how jinc_batch_process pins 512 distinct caller buffers (measurement).
usage: python profiles/probes/hostreg_probe.py"""
import ctypes as C
import json
import threading
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
hip.hipFree(None)


def one(nbytes, count, touch=True):
    bufs = [np.empty(nbytes, np.uint8) for _ in range(count)]
    if touch:
        for b in bufs:
            b[::4096] = 1
    t0 = time.perf_counter()
    for b in bufs:
        rc = hip.hipHostRegister(b.ctypes.data, nbytes, 0)
        assert rc == 0, rc
    t1 = time.perf_counter()
    for b in bufs:
        hip.hipHostUnregister(b.ctypes.data)
    t2 = time.perf_counter()
    return (t1 - t0) / count * 1e6, (t2 - t1) / count * 1e6


for mb in (2, 8, 32):
    reg, unreg = one(mb << 20, 32)
    print(json.dumps({"buffer_MiB": mb, "threads": 1, "register_us": round(reg, 1), "unregister_us": round(unreg, 1)}), flush=True)
reg, unreg = one(8 << 20, 16, touch=False)
print(json.dumps({"buffer_MiB": 8, "threads": 1, "untouched_pages": True, "register_us": round(reg, 1), "unregister_us": round(unreg, 1)}), flush=True)
for nt in (2, 4):
    res = [None] * nt

    def work(k):
        res[k] = one(8 << 20, 32)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    el = time.perf_counter() - t0
    print(json.dumps({"buffer_MiB": 8, "threads": nt, "register_us_per_thread": round(np.mean([r[0] for r in res]), 1),
                      "unregister_us_per_thread": round(np.mean([r[1] for r in res]), 1),
                      "pairs_per_s_all_threads": round(nt * 32 / el, 1)}), flush=True)
