import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from adamml_amd.hip import ConvDesc, call, ptr
from tools.bench_kernels import timeit
DEV="cuda"
for (cin,cout,n,h) in [(64,256,576,56),(256,64,576,56),(128,512,288,28)]:
    d=ConvDesc(n,h,h,cin,h,h,cout,1,1,1,0,1,1,0)
    x=torch.randn(n,h,h,cin,device=DEV).bfloat16(); w=torch.randn(cout,cin,device=DEV).bfloat16()
    y=torch.empty(n,h,h,cout,device=DEV,dtype=torch.bfloat16)
    stats=torch.zeros(32*2*cout,dtype=torch.float64,device=DEV)
    sc,sh=torch.rand(cin,device=DEV)+0.5, torch.randn(cin,device=DEV)
    by=2.0*n*h*h*(cin+cout)
    for name,(a,b,s) in {"transform+stats":(sc,sh,stats),"stats only":(None,None,stats),"transform only":(sc,sh,None),"plain":(None,None,None)}.items():
        t=timeit(lambda: call("adamml_conv_fwd",byref(d),ptr(x),ptr(w),ptr(a),ptr(b),ptr(y),ptr(s)),10)
        print("%4d->%4d %-16s %.3f ms  %.0f GB/s"%(cin,cout,name,t,by/t/1e6))
