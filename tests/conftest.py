"""Shared test plumbing.

Markers:  @pytest.mark.gpu  -> needs a real MI355X (run with `-m gpu` on the GPU box);
          everything else runs on CPU (`-m "not gpu"`), including the gloo world_size-2 tests.
The CPU oracle (oracle/) is the checker here and ONLY here (+ smoke() and bench.py's cpu_baseline).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_native_builds():
    """A fresh checkout has no binaries (they are git-ignored): build the product library with hipcc
    (cross-compiles for gfx950 without a GPU) before anything imports geot_amd.  This lives in the TEST
    harness on purpose - the package itself never builds or falls back, it fails loudly."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_geot_lib_bootstrap", os.path.join(ROOT, "geot_amd", "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.needs_build():
        mod.build()
    if mod.plugin_needs_build():
        mod.build_plugin()


_ensure_native_builds()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (HIP kernels are executed)")


def load_golden(fname):
    """{case: {array_name: ndarray}} from tests/golden/<fname>."""
    z = np.load(os.path.join(GOLDEN, fname))
    cases = {}
    for key in z.files:
        case, name = key.split("/", 1)
        cases.setdefault(case, {})[name] = z[key]
    return cases


def sorted_index(rng, nnz, keys, force_last=True):
    idx = np.sort(rng.integers(0, keys, nnz)).astype(np.int64)
    if force_last and nnz:
        idx[-1] = keys - 1
    return idx


def powerlaw_index(nnz, keys, seed):
    """SURVEY.md section 8d generator: w_k ~ rank^(-1/1.5), ranks randomly permuted, sorted draws."""
    rng = np.random.default_rng(seed)
    w = np.arange(1, keys + 1, dtype=np.float64) ** (-1.0 / 1.5)
    cdf = np.cumsum(w)
    perm = rng.permutation(keys)
    r = np.searchsorted(cdf, rng.random(nnz) * cdf[-1])
    idx = np.sort(perm[np.minimum(r, keys - 1)]).astype(np.int64)
    idx[-1] = keys - 1
    return idx


@pytest.fixture(scope="session")
def oracle():
    from oracle import api
    api.build()
    return api
