import sys, os, pickle, ctypes as C
import numpy as np
sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
c=pickle.load(open('/tmp/cyl_corpus.pkl','rb'))
big=[x for x in c if x[1].startswith('count') or float(x[1].split('dp=')[1])>1e-3 or float(x[1].split('dn=')[1].split()[0])>3e-3]
v=big[int(sys.argv[1])][0]
lib=oracle_lib.load(sys.argv[2]=='f32')
cc=v[0:3]; Rc=v[3:12]; rad=v[12]; hl=v[13]; cb=v[14:17]; Rb=v[17:26]; hb=v[26:29]
print('cc',cc,'a',Rc[[2,5,8]],'rad',rad,'hl',hl,'cb',cb,'hb',hb, 'Rb',Rb)
out=np.zeros(40)
arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
n=lib.pmgo_cyl_box(arr(cc),arr(Rc),C.c_double(rad),C.c_double(hl),arr(cb),arr(Rb),arr(hb),C.c_double(0.002),out.ctypes.data_as(C.c_void_p))
print(n,out[:10*n])
