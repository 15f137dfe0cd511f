import sys, os, pickle, ctypes as C
sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
task,kw,cases=pickle.load(open(sys.argv[1],'rb'))
L64=oracle_lib.load(False); L32=oracle_lib.load(True)
def ik(lib,q,t,it):
    out=np.zeros(9); tq=np.array([0,-1,0,0.])
    arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
    lib.pmgo_ik.restype=C.c_int
    n=lib.pmgo_ik(arr(q),arr(t),arr(tq),it,C.c_double(1e-5),out.ctypes.data_as(C.c_void_p))
    return n,out
for ci in [0,50,100]:
    t,i,s0,a,err=cases[ci]
    q=s0[0:9].astype(np.float64); ee=s0[18:21].astype(np.float64)
    lo=np.array([-0.67,-0.20,0.175]); hi=np.array([-0.37,0.20,0.55])
    tgt=np.clip((ee.astype(np.float32)+(a[:3]*np.float32(0.01)).astype(np.float32)).astype(np.float64),lo,hi)
    print('case',ci,'q',q[:7],'tgt',tgt)
    for it in (1,2,5,10,20,40):
        n64,o64=ik(L64,q,tgt,it); n32,o32=ik(L32,q,tgt,it)
        print(it, n64, n32, '%.2e'%np.abs(o64-o32)[:7].max(), 'step64 %.2e'%np.abs(o64-q)[:7].max())
