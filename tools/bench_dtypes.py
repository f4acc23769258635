import sys, torch
sys.path.insert(0, "/root/repo")
from bench import powerlaw_index, NNZ, KEYS
from geot_amd import hip
dev = torch.device("cuda")
index = powerlaw_index(NNZ, KEYS, 0, dev)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for F in (64, 128):
    for dt in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
        src = torch.rand(NNZ, F, device=dev).to(dt)
        out = torch.empty(KEYS, F, device=dev, dtype=dt)
        es = src.element_size()
        alg = NNZ * (es * F + 8) + KEYS * es * F
        for red in ("sum", "max", "mean"):
            t = timeit(lambda: hip.index_scatter_out(index, src, out, True, red))
            print(f"F={F:3d} {str(dt)[6:]:9s} {red:4s}: {t:.4f} ms  {NNZ / t / 1e6:6.2f} Gedge/s  {alg / t / 1e9:.2f} TB/s algorithmic ({alg / t / 1e9 / 8 * 100:.1f}% of 8 TB/s)")
        del src, out
