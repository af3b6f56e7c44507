import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from adamml_amd import hip
from adamml_amd.hip import ConvDesc, call, ptr
from tools.bench_kernels import timeit
DEV="cuda"
tot={"f":0,"d":0,"w":0}
for (n,h,c,s,cnt) in [(72,128,32,1,1),(72,128,96,2,1),(72,64,144,1,1),(72,64,144,2,1),(72,32,192,1,2),(72,32,192,2,1),(72,16,384,1,4),(72,16,576,1,2),(72,16,576,2,1),(72,8,960,1,3)]:
    oh=(h+2-3)//s+1
    d=ConvDesc(n,h,h,c,oh,oh,c,3,3,s,1,1,2,0)
    x=torch.randn(n,h,h,c,device=DEV).bfloat16(); w=torch.randn(9,c,device=DEV)
    y=torch.empty(n,oh,oh,c,device=DEV,dtype=torch.bfloat16); dz=torch.randn(n,oh,oh,c,device=DEV).bfloat16(); dx=torch.empty_like(x)
    dw=torch.zeros(c,1,3,3,device=DEV)
    stats=torch.zeros(32*2*c,dtype=torch.float64,device=DEV)
    sc,sh=torch.rand(c,device=DEV)+0.5, torch.randn(c,device=DEV)
    ws=hip.wgrad_workspace(d,0,x.device,depthwise=True)
    by=2.0*(x.numel()+y.numel())
    tf=timeit(lambda: call("adamml_dwconv_fwd",byref(d),ptr(x),ptr(w),ptr(sc),ptr(sh),ptr(y),ptr(stats)),10)
    td=timeit(lambda: call("adamml_dwconv_bwd_data",byref(d),ptr(dz),ptr(w),ptr(dx),0),10)
    tw=timeit(lambda: call("adamml_dwconv_bwd_weight",byref(d),ptr(dz),ptr(x),ptr(sc),ptr(sh),ptr(dw),ptr(ws),ws.numel()*4),10)
    tot["f"]+=tf*cnt; tot["d"]+=td*cnt; tot["w"]+=tw*cnt
    print("N%3d H%3d C%3d s%d x%d | fwd %.3f ms %5.0f GB/s | dgrad %.3f %5.0f | wgrad %.3f %5.0f"%(n,h,c,s,cnt,tf,by/tf/1e6,td,by/td/1e6,tw,by/tw/1e6))
print("TOTAL per MobileNet pass: fwd %.2f dgrad %.2f wgrad %.2f ms"%(tot["f"],tot["d"],tot["w"]))
