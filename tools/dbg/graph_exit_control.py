#!/usr/bin/env python3
"""Control for the exit-time SIGSEGV of hang_hunt's `graphs` scenario: the same capture / replay pattern with (a) torch ops only,
(b) geot ops, (c) geot ops + explicit teardown (del graphs, clear caches, release workspaces, synchronize) before exit."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["R"])
mode = os.environ["MODE"]
if mode != "torch":
    import geot_amd as geot
    from geot_amd import hip, ops
    ops.set_option("slab_mode", "always")
nodes, nnz, H, F = 60_000, 12_000_000, 4, 64
gs, keep = [], []
for i in range(2):
    g = torch.Generator(device="cuda"); g.manual_seed(i)
    di = torch.randint(0, nodes, (nnz,), device="cuda", generator=g).sort().values; di[-1] = nodes - 1
    si = torch.randint(0, nodes, (nnz,), device="cuda", generator=g)
    w = torch.rand(nnz, H, device="cuda", generator=g)
    x = torch.rand(nodes, H, F, device="cuda", generator=g)
    f = (lambda: geot.mh_spmm(si, di, w, x)) if mode != "torch" else (lambda: torch.zeros(nodes, H * F, device="cuda").index_add_(0, di[:1000000], x.view(nodes, -1)[si[:1000000]]))
    for _ in range(3): y = f()
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        y = f()
    gs.append(cg); keep.append((si, di, w, x, y))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for _ in range(10):
    for cg, s in zip(gs, streams):
        with torch.cuda.stream(s): cg.replay()
    torch.cuda.synchronize()
if mode == "geot_teardown":
    del gs, cg, y, keep
    ops.clear_caches(); hip.release_workspaces()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
print("DONE", flush=True)
'''
for mode in ("torch", "geot", "geot_teardown"):
    rcs = []
    for i in range(12):
        p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MODE=mode, R=ROOT), capture_output=True, text=True, timeout=180)
        rcs.append(p.returncode if "DONE" in p.stdout else ("noDONE", p.returncode))
    print(mode, rcs, flush=True)
