import torch, time
dev="cuda"
def t(fn, it=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e-3
for mb in (8,16,32,64,96,128,192,256,512,1024,2048):
    n=mb*2**20//4
    x=torch.randn(n,device=dev); y=torch.empty_like(x)
    s=t(lambda: y.copy_(x))
    s2=t(lambda: x.mul_(1.0001))
    s3=t(lambda: y.fill_(1.0))
    print("buffers of %5d MB: copy (r+w) %6.2f TB/s | in-place scale (r+w same buffer) %6.2f TB/s | fill (w) %6.2f TB/s" % (mb, 2*mb*2**20/s/1e12, 2*mb*2**20/s2/1e12, mb*2**20/s3/1e12))
