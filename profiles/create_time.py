import sys, time
sys.path.insert(0,'.')
import torch
import __graft_entry__ as g
pkg=g.load_package()
for name,(fmt,sw,sh,dw,dh,kw) in {"warm":("Y8",64,64,128,128,{}),"C2":("Y8",1920,1080,3840,2160,dict(tap=3)),"C3":("YUV420P16",1920,1080,3840,2160,dict(tap=8)),"C4":("RGBPS",3840,2160,7680,4320,dict(tap=4,blur=0.98)),"N15":("Y8",1280,720,1920,1080,{}),"N3":("Y8",1280,720,3840,2160,{}),"A137":("Y8",1280,720,1754,986,{}),"T16":("Y8",1920,1080,3840,2160,dict(tap=16)),"D12":("Y8",3840,2160,1920,1080,{})}.items():
    t0=time.perf_counter()
    f=pkg.Filter(pkg.FORMATS[fmt],sw,sh,dw,dh,device=0,**kw)
    t1=time.perf_counter()
    f.close()
    print(f"{name:5s} create+upload {1e3*(t1-t0):8.1f} ms")
