import os, sys, ctypes, numpy as np, torch
sys.path.insert(0,'/root/repo')
import suffix_amd
import _devlib
from suffix_amd import device as sdev
from suffix_amd.device import _p
eng = _devlib.engine()
z = np.load('/root/repo/tests/golden/fasta_fixtures.npz')
host = np.ascontiguousarray(z['AP009048_10000'])
text = torch.from_numpy(host).cuda()
n = host.size
ws = sdev.sa_workspace(n, text.device)
sa = torch.empty(n, dtype=torch.int32, device='cuda')
for _ in range(3): sdev.build_sa(text, out=sa, workspace=ws)
torch.cuda.synchronize()
w = ws[:64].cpu().numpy().view(np.uint32)
print('status', w[0], 'phases (100 MHz ticks):', w[2:10].tolist())
