import sys, time, torch, subprocess, os
sys.path.insert(0, "/root/repo")
ROOT = "/root/repo"
def wall(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
if len(sys.argv) > 1 and sys.argv[1] == "shim":
    torch.ops.load_library(os.path.join(ROOT, "geot_amd", "_C.so"))
    idx = torch.sort(torch.randint(0, 10000, (100000,), device="cuda")).values; idx[-1] = 9999
    src = torch.rand(100000, 32, device="cuda")
    print(f"C++ shim  torch.ops.geot.index_scatter (item sync + empty + 2 kernels): {wall(lambda: torch.ops.geot.index_scatter(0, idx, src, 'sum', True)):.1f} us/call")
    ref = lambda: torch.zeros(10000, 32, device='cuda').index_add_(0, idx, src)
    print(f"torch zeros+index_add_ (for scale): {wall(ref):.1f} us/call")
    sys.exit()
import geot_amd as geot
from geot_amd import hip
idx = torch.sort(torch.randint(0, 10000, (100000,), device="cuda")).values; idx[-1] = 9999
src = torch.rand(100000, 32, device="cuda")
out = torch.empty(10000, 32, device="cuda")
print(f"python op geot.index_scatter (row rule + empty + 2 kernels): {wall(lambda: geot.index_scatter(0, src, idx)):.1f} us/call")
print(f"python doorway hip.index_scatter_out (no row rule):  {wall(lambda: hip.index_scatter_out(idx, src, out)):.1f} us/call")
os.environ["X"]="1"
subprocess.run([sys.executable, __file__, "shim"])
print(f"python op geot.index_scatter sorted=False (probe + empty + 2 kernels): {wall(lambda: geot.index_scatter(0, src, idx, 'sum', False)):.1f} us/call")
