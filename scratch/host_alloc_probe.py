import numpy as np, time, os, sys, ctypes
def T(f, *a):
    t=time.perf_counter(); r=f(*a); return r, (time.perf_counter()-t)*1e3
n=1_000_000
apps=np.ones(n,dtype=np.int64)
def probe(tag):
    for rep in range(2):
        c,t1=T(lambda: np.cumsum(apps,dtype=np.int64))
        r,t2=T(lambda: c/float(n))
        r2,t3=T(lambda: r*100)
        b,t5=T(lambda: np.empty(64<<20,dtype=np.uint8))
        _,t6=T(lambda: b.fill(1))
        _,t7=T(lambda: b.fill(2))
        del b
        print(f"{tag}: cumsum {t1:.2f} div {t2:.2f} mul(new) {t3:.2f} fill-first64M {t6:.2f} fill-again {t7:.2f} ms", flush=True)
probe("before gpu init")
import torch
x=torch.ones(n,dtype=torch.int64,device='cuda'); torch.cuda.synchronize()
probe("after gpu init")
big=torch.empty(1<<30,dtype=torch.uint8,device='cuda'); torch.cuda.synchronize()
probe("after 1 GiB device alloc")
pin=torch.empty(64<<20,dtype=torch.uint8).pin_memory()
probe("after pinned alloc")
if len(sys.argv)>1:
    libc=ctypes.CDLL("libc.so.6")
    libc.mallopt(-3, 1<<30)   # M_MMAP_THRESHOLD
    libc.mallopt(-1, 1<<31 - 1)   # M_TRIM_THRESHOLD
    probe("mallopt no-mmap")
    probe("mallopt no-mmap again")
