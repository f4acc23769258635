import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import powerlaw_index
import geot_amd as geot
from geot_amd import hip
nnz, keys, F = 10_000_000, 1_000_000, 64
index = torch.from_numpy(powerlaw_index(nnz, keys, 0)).cuda()
torch.manual_seed(1)
src = torch.rand(nnz, F, device="cuda")
ref = torch.zeros(keys, F, device="cuda", dtype=torch.float64).index_add_(0, index, src.double())
for mode in (1, 0):
    hip.set_option("handoff", mode)
    out = geot.index_scatter(0, src, index, "sum", True)
    rel = ((out.double() - ref).abs() / (ref.abs() + 1e-30))
    cs = (out.double().sum(0) - src.double().sum(0)).abs() / src.double().sum(0)
    print("handoff", mode, "max rel err", rel.max().item(), "rows > 1e-6:", int((rel.max(1).values > 1e-6).sum()), "checksum rel err max", cs.max().item())
