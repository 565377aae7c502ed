import torch, time
from newsreclib_amd import ops
torch.manual_seed(0)
n=211200; V=70000
for name, ids in [("uniform+30%pad", torch.where(torch.rand(n)<0.3, torch.zeros(n,dtype=torch.long), torch.randint(1,V,(n,)))),
                  ("zipf", (torch.distributions.Pareto(1.0,1.1).sample((n,)).long().clamp(max=V-1)))]:
    ids=ids.cuda()
    for _ in range(3): o=ops.sort_positions(ids,V)
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): o=ops.sort_positions(ids,V)
    b.record(); torch.cuda.synchronize()
    s=ids[o]; assert bool((s[1:]>=s[:-1]).all()) and o.sort().values.equal(torch.arange(n,device='cuda'))
    print(name, a.elapsed_time(b)/20*1000, "us per sort")
