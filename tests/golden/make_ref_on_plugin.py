#!/usr/bin/env python3
"""Build-container script (needs /root/reference; never runs on the GPU box): loads geot_amd/_C.so and imports the
reference's UNMODIFIED Python wrappers geot/gather_scatter.py, geot/gather_weight_scatter.py, geot/csr_gws.py on top of
it - in both orders (plugin first, as a test harness would; wrappers first, as the reference's geot/__init__.py:4-19
does) - each in a fresh process, and records what the dispatcher then holds in tests/golden/ref_wrappers_on_plugin.json.

The wrappers are imported by path with importlib (`import geot` of the reference needs a working
Triton).  Nothing of the reference is copied: the fixture holds schema strings, dispatch-key names, one error text of
OUR plugin and shapes.

    python tests/golden/make_ref_on_plugin.py            # rewrites the fixture
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PLUGIN = os.path.join(ROOT, "geot_amd", "_C.so")
REF = "/root/reference/geot"
OUT = os.path.join(HERE, "ref_wrappers_on_plugin.json")
WRAPPERS = ("gather_scatter", "gather_weight_scatter", "csr_gws")

PROBE = r'''
import importlib.util, json, sys, torch
plugin, ref, order = sys.argv[1:4]
names = ("gather_scatter", "gather_weight_scatter", "csr_gws")

def load_wrappers():
    mods = {}
    for n in names:
        spec = importlib.util.spec_from_file_location("refgeot_" + n, f"{ref}/{n}.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[n] = m
    return mods

def has_schema(q):
    try:
        torch._C._dispatch_find_schema_or_throw(q, "")
        return True
    except RuntimeError:
        return False

res = {"order": order}
if order == "plugin_first":
    torch.ops.load_library(plugin)
    every = sorted(n for n in torch._C._dispatch_get_all_op_names() if n.startswith("geot::"))
    res["plugin_defines"] = [n for n in every if has_schema(n)]
    res["plugin_implements_only"] = [n for n in every if not has_schema(n)]
    mods = load_wrappers()
else:
    mods = load_wrappers()
    torch.ops.load_library(plugin)
res["import_error"] = None
res["schemas"] = {n: str(getattr(torch.ops.geot, n).default._schema) for n in names}
# who serves which dispatch key: the plugin's C++ kernels (torch_ops.cpp) or the wrapper's Python (custom_ops.py)
table = {}
for n in names:
    keys = {}
    for line in torch._C._dispatch_dump("geot::" + n).splitlines():
        if ": registered at " in line and not line.startswith(("debug", "schema", "name")):
            key, rest = line.split(": registered at ", 1)
            keys[key.replace("[alias]", "")] = "plugin" if "torch_ops.cpp" in rest else ("python" if "custom_ops.py" in rest else rest.split(" ")[0])
    table[n] = keys
res["dispatch"] = table
# a call with CPU tensors lands in the plugin (its refusal text), through the reference's own Python function
x = torch.rand(4, 3)
i = torch.tensor([0, 1])
errs = {}
for n, args in (("gather_scatter", (i, i, x)), ("gather_weight_scatter", (i, i, torch.rand(2), x)),
                ("csr_gws", (torch.tensor([0, 1, 2, 2, 2]), i, torch.rand(2), x))):
    try:
        getattr(mods[n], n)(*args)
        errs[n] = None
    except Exception as e:
        errs[n] = f"{type(e).__name__}: {str(e).splitlines()[0]}"
res["cpu_call"] = errs
# the reference's fake-tensor rules on the plugin's ops (geot/gather_scatter.py:12-18): [dynamic rows, F]
from torch._subclasses.fake_tensor import FakeTensorMode
from torch.fx.experimental.symbolic_shapes import ShapeEnv
fake = {}
with FakeTensorMode(shape_env=ShapeEnv(), allow_non_fake_inputs=False) as mode:
    fx, fi, fw = mode.from_tensor(x), mode.from_tensor(i), mode.from_tensor(torch.rand(2))
    for n, args in (("gather_scatter", (fi, fi, fx)), ("gather_weight_scatter", (fi, fi, fw, fx)), ("csr_gws", (fi, fi, fw, fx))):
        y = getattr(torch.ops.geot, n)(*args)
        fake[n] = {"ndim": y.dim(), "cols": int(y.shape[1]), "rows_is_symbolic": not isinstance(y.shape[0], int), "dtype": str(y.dtype)}
res["fake"] = fake
print("RESULT " + json.dumps(res))
'''


def run(order):
    p = subprocess.run([sys.executable, "-c", PROBE, PLUGIN, REF, order], capture_output=True, text=True, timeout=600, cwd="/tmp")
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    if p.returncode != 0 or not lines:
        return {"order": order, "import_error": (p.stderr or p.stdout)[-2000:]}
    return json.loads(lines[-1][7:])


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    import torch
    res = {"made_by": "tests/golden/make_ref_on_plugin.py", "torch": torch.__version__,
           "wrappers": [f"geot/{n}.py" for n in WRAPPERS], "runs": [run("plugin_first"), run("wrappers_first")]}
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
