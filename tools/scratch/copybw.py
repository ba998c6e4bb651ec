"""Practical HBM copy / read / write bandwidth of torch elementwise kernels at several sizes (the yardstick for the
HBM-bound kernels: 33.5 MB in + 33.5 MB out copies in ~10 us on an MI355X)."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(True); e1=torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for mb in (8, 16, 33.5, 67, 134, 268, 1072):
    n=int(mb*1e6/4); x=torch.randn(n,device="cuda"); y=torch.empty_like(x)
    us=t(lambda: y.copy_(x)); us2=t(lambda: torch.mul(x, 2.0, out=y)); us3=t(lambda: y.fill_(1.0)); us4=t(lambda: x.sum())
    print("%7.1f MB copy %6.1f us %.2f TB/s | mul %6.1f us %.2f TB/s | fill %6.1f us %.2f TB/s(w) | sum %6.1f us %.2f TB/s(r)"%(mb, us, 2*mb/us, us2, 2*mb/us2, us3, mb/us3, us4, mb/us4))
