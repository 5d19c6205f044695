import sys, ctypes as C
sys.path.insert(0,'/root/repo')
from outfit_amd import _lib
L=_lib.lib()
for n,seed in ((1<<34, 7),(1<<34, 123456789)):
    c=(C.c_uint64*4)()
    rc=L.outfit_hip_selftest_crmath(n, seed, c)
    print("n",n,"rc",rc,"counts",list(c),"slow fraction",c[2]/n)
