import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from newsreclib_amd.trainer import FlatParams, FusedAdam, LazyTableAdam
DEV = "cuda"
V, D = 1000, 300
gen = torch.Generator().manual_seed(11)
def make():
    torch.manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(V, D, device=DEV)), torch.nn.Parameter(torch.randn(129, device=DEV))]
    flat = FlatParams(ps)
    return ps, flat, FusedAdam(flat, lr=1e-3)
pa, fa, oa = make(); pb, fb, ob = make()
lazy = LazyTableAdam(fb, ob, pb[0], period=64)
for t in range(1, 6):
    ids = torch.randint(0, V, (50,), generator=gen).to(DEV)
    uniq = torch.unique(ids)
    g_rows = torch.randn(uniq.numel(), D, generator=gen).to(DEV)
    before = pa[0].detach()[uniq].clone()
    lazy.begin(ids.reshape(-1, 1), None)
    torch.cuda.synchronize()
    d = (pb[0].detach()[uniq] - before).abs()
    bad = (d.max(1).values > 0).nonzero().flatten()
    print("step", t, "gather rows differ:", bad.numel(), "of", uniq.numel(), "max", float(d.max()), "last of bad rows", lazy.last[uniq[bad]].tolist()[:8])
    for f in (fa, fb):
        f.grad[: V * D].view(V, D)[uniq] = g_rows
    oa.step(grad_scale=1.0, zero_grad=True)
    lazy.finish(1.0)
    torch.cuda.synchronize()
    # rows updated this step
    d2 = (pb[0].detach()[uniq] - pa[0].detach()[uniq]).abs()
    dm = (ob.exp_avg[:V*D].view(V, D)[uniq] - oa.exp_avg[:V*D].view(V, D)[uniq]).abs()
    dv = (ob.exp_avg_sq[:V*D].view(V, D)[uniq] - oa.exp_avg_sq[:V*D].view(V, D)[uniq]).abs()
    print("   after update: p", float(d2.max()), "m", float(dm.max()), "v", float(dv.max()))
lazy.flush(); torch.cuda.synchronize()
print("final", float((fb.flat - fa.flat).abs().max()), float((ob.exp_avg - oa.exp_avg).abs().max()), float((ob.exp_avg_sq - oa.exp_avg_sq).abs().max()))
d = (fb.flat - fa.flat)[:V*D].view(V, D).abs().max(1).values
print("rows differing", int((d > 0).sum()))
