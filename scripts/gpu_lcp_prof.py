import sys, json, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import _gen, suffix_amd
from suffix_amd import device as sdev
eng=suffix_amd.default_engine(); dev=torch.device('cuda',0)
for name,gen in (("dna1g", lambda: _gen.dna(1_000_000_000, seed=7)), ("eng400m", lambda: _gen.english_like(400_000_000, seed=3))):
    t=torch.from_numpy(gen()).to(dev); n=t.numel()
    sa=sdev.build_sa(t); torch.cuda.synchronize()
    eng.profile(True); eng.profile_reset()
    lcp=sdev.build_lcp(t, sa); torch.cuda.synchronize()
    print(name, n, {r["name"]: round(r["total_ms"],2) for r in eng.profile_report()})
    eng.profile_reset()
    sdev.build_sa(t); torch.cuda.synchronize()
    print(name, "SA", {r["name"]: round(r["total_ms"],2) for r in eng.profile_report()}, eng.build_stats())
    eng.profile(False)
    del t, sa, lcp
