import torch,time
n=3600*1024*1024
x=torch.empty(n,dtype=torch.uint8,device='cuda')
y=torch.empty(n//2,dtype=torch.uint8,device='cuda')
def t(f,k=5):
    f(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/k
ms=t(lambda: x.zero_()); print("fill 3.6 GiB: %.3f ms -> %.2f TB/s"%(ms, n/ms/1e9))
ms=t(lambda: x[:n//2].copy_(y)); print("copy 1.8 GiB -> 1.8 GiB: %.3f ms -> %.2f TB/s (r+w)"%(ms, n/ms/1e9))
xi=x.view(torch.int32)
ms=t(lambda: xi.sum()); print("read 3.6 GiB (sum): %.3f ms -> %.2f TB/s"%(ms, n/ms/1e9))
