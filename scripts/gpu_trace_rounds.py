import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import _gen, suffix_amd
import _devlib
from suffix_amd import device as sdev
eng = _devlib.engine()
dev = torch.device('cuda', 0)
name = sys.argv[1]
gen = {'c5': _gen.utf8_mixed, 'c3': _gen.english_like, 'dup': _gen.near_duplicates}[name]
t = torch.from_numpy(gen(1_000_000_000)).to(dev)
ws = sdev.sa_workspace(t.numel(), dev)
sa = torch.empty(t.numel(), dtype=torch.int32, device=dev)
sdev.build_sa(t, out=sa, workspace=ws); torch.cuda.synchronize()
t0 = time.perf_counter(); sdev.build_sa(t, out=sa, workspace=ws); torch.cuda.synchronize()
print(name, 'ms', (time.perf_counter() - t0) * 1e3, eng.build_stats(), file=sys.stderr)
