"""What a gradient all-reduce costs inside a compute loop on this stack (1-rank RCCL group is enough to see the stream
hand-over cost): back-to-back collectives vs collectives interleaved with kernels on torch's current stream."""
import os, sys, time
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.randn(233000, device="cuda")
a = torch.randn(2048, 2048, device="cuda")
for _ in range(5): dist.all_reduce(x)
torch.cuda.synchronize()

def loop(tag, coll, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        b = a @ a
        h0 = time.perf_counter(); coll(); host += time.perf_counter() - h0
        x.mul_(1.0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag:34s}: {1e3*(t2-t0)/n:8.3f} ms per iteration (host in the collective {1e6*host/n:8.1f} us, host loop {1e3*(t1-t0)/n:.3f} ms)", flush=True)

loop("no collective", lambda: None)
loop("all_reduce (sync op)", lambda: dist.all_reduce(x))
loop("all_reduce async_op + wait", lambda: dist.all_reduce(x, async_op=True).wait())
side = torch.cuda.Stream()
def on_side():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(x)
    torch.cuda.current_stream().wait_stream(side)
loop("all_reduce issued from a side stream", on_side)
loop("no collective (again)", lambda: None)
dist.destroy_process_group()
