import torch, time
dev=torch.device("cuda:0")
N,C=50000,10000
out=torch.empty(N,C,device=dev)
a=torch.randn(N,C,device=dev)
def t(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
ms=t(lambda: out.zero_()); print("zero_ 2GB: %.3f ms %.2f TB/s"%(ms, 2e9/ms/1e9))
ms=t(lambda: out.fill_(1.5)); print("fill_ 2GB: %.3f ms %.2f TB/s"%(ms, 2e9/ms/1e9))
ms=t(lambda: torch.add(a,1.0,out=out)); print("add r+w 4GB: %.3f ms %.2f TB/s"%(ms, 4e9/ms/1e9))
ms=t(lambda: out.copy_(a)); print("copy r+w 4GB: %.3f ms %.2f TB/s"%(ms, 4e9/ms/1e9))
