import sys, torch, numpy as np
sys.path.insert(0, ".")
import blah2_amd as b2
n=2_000_000; B=16
sa=b2.SpectrumAnalyser(n,2000,max_batch=B)
dev=torch.device("cuda",0)
x=torch.randn((B,n),dtype=torch.complex64,device=dev)
out=torch.zeros((B,2000),dtype=torch.complex128,device=dev)
st=torch.cuda.current_stream().cuda_stream
for b in (1,B):
    for _ in range(3): sa.process_dev(b2.FMT_C32,x.data_ptr(),b,n,out.data_ptr(),st)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): sa.process_dev(b2.FMT_C32,x.data_ptr(),b,n,out.data_ptr(),st)
    e1.record(); torch.cuda.synchronize()
    print("batch",b,"us per CPI",e0.elapsed_time(e1)/20/b*1e3)
