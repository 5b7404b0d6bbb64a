import torch, time
dev="cuda:0"
for mb in (23.66, 400, 1514):
    n=int(mb*1e6/2/4)
    a=torch.rand(n,device=dev); b=torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    reps=200 if mb<100 else 20
    s.record()
    for _ in range(reps): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    us=s.elapsed_time(e)*1e3/reps
    print("copy of %.1f MB total traffic: %.2f us per launch, %.0f GB/s" % (mb, us, mb*1e6/us/1e3))
