import torch, time
dev="cuda"
B=40960
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1000
for (i,o) in ((128,128),(96,128),(100,128),(128,12),(24,64),(64,20),(128,1),(120,20),(76,30)):
    x=torch.randn(B,i,device=dev); g=torch.randn(B,o,device=dev); w=torch.randn(o,i,device=dev)
    t_fwd=timeit(lambda: torch.nn.functional.linear(x,w))
    t_dx=timeit(lambda: g@w)
    t_dw=timeit(lambda: g.t()@x)
    res=[]
    for S in (16,32,64,128):
        def f():
            return torch.bmm(g.view(S,B//S,o).transpose(1,2), x.view(S,B//S,i)).sum(0)
        res.append((S,round(timeit(f),1)))
    ref=g.t()@x; alt=torch.bmm(g.view(64,B//64,o).transpose(1,2), x.view(64,B//64,i)).sum(0)
    print(f"in {i} out {o}: fwd {t_fwd:.1f} dx {t_dx:.1f} dw {t_dw:.1f} us | splitK {res} | maxerr {(ref-alt).abs().max().item():.2e} rel {((ref-alt).abs().max()/ref.abs().max()).item():.1e}")
