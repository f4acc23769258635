"""Loader / builder for libgeot_hip.so, the C-ABI HIP library declared in include/geot_hip.h.

The library is the product: there is NO fallback.  If it is missing or does not export the ABI
the import of :mod:`geot_amd` fails loudly (the reference does the same for its ``_C`` module,
geot/__init__.py:12-19: ``ImportError("Could not find module '_C' in ...")``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_NAME = "libgeot_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
# The DEVELOPMENT build: the same sources with -DGEOT_DEV_EXPERIMENTS (measured-and-rejected kernel variants and the timing probe
# behind geot_set_option switches; include/geot_hip_dev.h).  Loaded INSTEAD of the product library by a process started with
# GEOT_HIP_LIB=dev (tools/, the A/B tests); it carries the product's soname, so the torch plugin, which names libgeot_hip.so,
# binds to the copy already in the process.  Nothing that imports geot_amd without that variable ever sees it.
DEV_LIB_NAME = "libgeot_hip_dev.so"
DEV_LIB_PATH = os.path.join(_HERE, DEV_LIB_NAME)
DEV = os.environ.get("GEOT_HIP_LIB", "") == "dev"
# (the longest compiles first; seg_reduce_<type>.hip are seg_reduce.hip's kernels of one storage type each - they #include it)
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("seg_reduce_f32.hip", "seg_plan.hip", "seg_reduce_f16.hip", "seg_reduce_bf16.hip",
                                                     "seg_reduce_f64.hip", "seg_slab.hip", "seg_reduce.hip", "seg_sort.hip",
                                                     "seg_guard.hip")]
SEG_REDUCE = os.path.join(_HERE, "csrc", "seg_reduce.hip")
HEADER = os.path.join(_ROOT, "include", "geot_hip.h")
DEV_HEADER = os.path.join(_ROOT, "include", "geot_hip_dev.h")     # measurement hooks + experiment switches (not the stable ABI)
PLUGIN_PATH = os.path.join(_HERE, "_C.so")                       # the torch dispatcher plugin (csrc/torch_ops.cpp)
PLUGIN_SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("torch_ops.cpp", "host_state.cpp", "host_cache.cpp", "host_plan.cpp")]
PLUGIN_HEADER = os.path.join(_HERE, "csrc", "host.h")
LIB_INPUTS = SOURCES + [HEADER, DEV_HEADER, os.path.join(_HERE, "csrc", "internal.h")]
PLUGIN_INPUTS = PLUGIN_SOURCES + [PLUGIN_HEADER, HEADER]

GEOT_OK = 0
GEOT_F32, GEOT_F64, GEOT_F16, GEOT_BF16 = 0, 1, 2, 3
GEOT_W_EDGE_MAJOR, GEOT_W_HEAD_MAJOR = 0, 1
ABI_VERSION = 2

#: every symbol include/geot_hip.h and include/geot_hip_dev.h declare (tests check the library exports all of them)
SYMBOLS = [
    "geot_abi_version", "geot_last_error", "geot_build_info", "geot_workspace_bytes", "geot_mh_workspace_bytes",
    "geot_workspace_init", "geot_index_scatter", "geot_index_scatter_reduce", "geot_gather_reduce", "geot_gather_scatter",
    "geot_gather_weight_scatter", "geot_mh_spmm", "geot_sddmm_coo", "geot_mh_sddmm_coo", "geot_gather_select_backward", "geot_gather_rows", "geot_index_probe",
    "geot_publish_word", "geot_publish_pending", "geot_set_alarm_word", "geot_content_fingerprint", "geot_content_fingerprint_scratch_bytes", "geot_index_probe_range", "geot_sort_supported", "geot_sort_workspace_bytes", "geot_sort_index",
    "geot_csr_workspace_bytes", "geot_csr_gws", "geot_coo_to_csr",
    "geot_slab_units", "geot_slab_full_chip", "geot_slab_rows_per_group", "geot_slab_rows_per_group_dtype", "geot_slab_units_for", "geot_slab_rows_per_group_shape", "geot_slab_workspace_bytes", "geot_slab_workspace_bytes_staged", "geot_slab_spmm", "geot_slab_sddmm", "geot_slab_sddmm_staged", "geot_slab_mh_sddmm", "geot_slab_to_plan_order",
    "geot_slab_plan_scratch_bytes", "geot_slab_plan_rows", "geot_slab_plan_groups", "geot_slab_plan_edges",
    "geot_profile_enable", "geot_profile_reset", "geot_profile_read", "geot_profile_box", "geot_profile_box_rows", "geot_tune", "geot_set_option", "geot_last_kernel",
]

class SlabPlan(ctypes.Structure):
    """geot_slab_plan of include/geot_hip.h (device pointers as integers)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("e_src", "e_dl", "e_perm", "g_begin", "g_vrow0", "g_nv", "v_out",
                                                "c_row", "c_first", "c_count", "v_row", "v_total", "c_total")] + \
               [(n, ctypes.c_int64) for n in ("n_groups", "n_vrows", "n_carry", "n_split", "nnz")] + \
               [("units", ctypes.c_int32), ("rows_per_group", ctypes.c_int32), ("slab_shift", ctypes.c_int32),
                ("n_slabs", ctypes.c_int32)]


_lib = None


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths) -> str:
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(binary: str, sources) -> bool:
    """Is `binary` older than its sources?  By CONTENT where a stamp (written next to the binary by the build) exists - file
    times do not survive every copy of the tree (the GPU box gets a snapshot), and a spurious rebuild costs minutes - else by
    modification time."""
    if not os.path.exists(binary):
        return True
    stamp = binary + ".srchash"
    if os.path.exists(stamp):
        with open(stamp) as f:
            return f.read().strip() != _digest(sources)
    t = os.path.getmtime(binary)
    return any(os.path.getmtime(p) > t for p in sources)


def stamp(which=None) -> None:
    """Record the content of the sources a binary was built from.  `which`: LIB_PATH, DEV_LIB_PATH or PLUGIN_PATH - the one just
    built; None (`make shim`'s g++ recipe, which builds behind this module's back) stamps whichever exists."""
    for binary, sources in ((LIB_PATH, LIB_INPUTS), (DEV_LIB_PATH, LIB_INPUTS), (PLUGIN_PATH, PLUGIN_INPUTS)):
        if os.path.exists(binary) and which in (None, binary):
            with open(binary + ".srchash", "w") as f:
                f.write(_digest(sources) + "\n")


def needs_build(dev: bool = False) -> bool:
    return _stale(DEV_LIB_PATH if dev else LIB_PATH, LIB_INPUTS)


OBJ_DIR = os.path.join(_HERE, "csrc", ".obj")        # per-source objects (git-ignored; they do not travel with a snapshot)
HIPFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I", os.path.join(_ROOT, "include")]


def build(force: bool = False, verbose: bool = False, jobs: int = 0, dev: bool = False) -> str:
    """Cross-compile the HIP library for gfx950 in-tree (works without a GPU).  One object per source, the stale ones compiled
    side by side (the tile kernel's instantiations are one object per storage type: the longest, fp32, ~1 minute), then one link.
    dev=True: the development build (DEV_LIB_PATH, -DGEOT_DEV_EXPERIMENTS, its own objects)."""
    lib_path = DEV_LIB_PATH if dev else LIB_PATH
    obj_dir = os.path.join(OBJ_DIR, "dev") if dev else OBJ_DIR
    flags = HIPFLAGS + (["-DGEOT_DEV_EXPERIMENTS"] if dev else [])
    if force or needs_build(dev):
        from concurrent.futures import ThreadPoolExecutor
        os.makedirs(obj_dir, exist_ok=True)
        shared = LIB_INPUTS[len(SOURCES):]
        todo, objs = [], []
        def inputs(src):
            part = os.path.basename(src).startswith("seg_reduce_")
            return [src] + ([SEG_REDUCE] if part else []) + shared

        for src in SOURCES:
            obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
            objs.append(obj)
            if force or _stale(obj, inputs(src)) or not os.path.exists(obj + ".srchash"):
                todo.append((src, obj))

        def compile_one(job):
            src, obj = job
            cmd = [hipcc(), *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(obj + ".srchash", "w") as f:
                f.write(_digest(inputs(src)) + "\n")

        with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, todo))
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,-soname,{LIB_NAME}", *objs, "-o", lib_path]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        stamp(lib_path)
    return lib_path


def plugin_needs_build() -> bool:
    return _stale(PLUGIN_PATH, PLUGIN_INPUTS)


def build_plugin(force: bool = False, verbose: bool = False) -> str:
    """g++ build of geot_amd/_C.so against the installed torch (no GPU needed): one object per source (csrc/torch_ops.cpp +
    csrc/host_*.cpp, compiled side by side, ~40 s), one link.  Same recipe as `make shim`."""
    if force or plugin_needs_build():
        import torch
        from concurrent.futures import ThreadPoolExecutor
        tdir = os.path.dirname(torch.__file__)
        os.makedirs(OBJ_DIR, exist_ok=True)
        flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
                 f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I", os.path.join(_ROOT, "include"),
                 "-I", os.path.join(tdir, "include"), "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
                 "-I", "/opt/rocm/include"]
        shared = [PLUGIN_HEADER, HEADER]
        todo, objs = [], []
        for src in PLUGIN_SOURCES:
            obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
            objs.append(obj)
            if force or _stale(obj, [src] + shared) or not os.path.exists(obj + ".srchash"):
                todo.append((src, obj))

        def compile_one(job):
            src, obj = job
            cmd = ["g++", *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(obj + ".srchash", "w") as f:
                f.write(_digest([src] + shared) + "\n")

        with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, todo))
        cmd = ["g++", "-shared", "-fPIC", *objs, "-o", PLUGIN_PATH, "-L", _HERE, "-lgeot_hip",
               "-L", os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
               "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{os.path.join(tdir, 'lib')}"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        stamp(PLUGIN_PATH)
    return PLUGIN_PATH


def load() -> ctypes.CDLL:
    """dlopen the library and attach prototypes.  Raises ImportError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = DEV_LIB_PATH if DEV else LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"Could not find '{os.path.basename(path)}' in {_HERE}: build it with `make {'devlib' if DEV else 'lib'}` (or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`).  geot_amd has no fallback path.")
    # (RTLD_GLOBAL + the product's soname: the torch plugin's DT_NEEDED libgeot_hip.so binds to THIS copy, product or development)
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{path} does not export {missing}")
    c_i64, c_int, c_vp, c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    L.geot_abi_version.restype = c_int
    L.geot_last_error.restype = ctypes.c_char_p
    L.geot_build_info.restype = ctypes.c_char_p
    L.geot_workspace_bytes.restype = c_sz
    L.geot_workspace_bytes.argtypes = [c_i64, c_i64, c_i64, c_int]
    L.geot_mh_workspace_bytes.restype = c_sz
    L.geot_mh_workspace_bytes.argtypes = [c_i64, c_i64, c_i64, c_i64, c_int]
    L.geot_workspace_init.argtypes = [c_vp, c_sz, c_vp]
    L.geot_index_probe.argtypes = [c_vp, c_i64, c_vp, c_vp]
    L.geot_index_probe_range.argtypes = [c_vp, c_i64, c_vp, c_vp]
    L.geot_publish_word.argtypes = [c_vp, c_vp, c_i64]
    L.geot_publish_pending.argtypes = []
    L.geot_set_alarm_word.argtypes = [c_vp]
    L.geot_content_fingerprint.argtypes = [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_vp, c_vp]
    L.geot_content_fingerprint_scratch_bytes.restype = c_sz
    L.geot_sort_supported.argtypes = [c_i64, c_i64, c_i64]
    L.geot_sort_workspace_bytes.restype = c_sz
    L.geot_sort_workspace_bytes.argtypes = [c_i64]
    L.geot_sort_index.argtypes = [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]
    L.geot_index_scatter.argtypes = [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_sz, c_vp]
    L.geot_index_scatter_reduce.argtypes = [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_sz, c_vp]
    L.geot_gather_reduce.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_sz, c_vp]
    L.geot_gather_scatter.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_gather_weight_scatter.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_mh_spmm.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_sz, c_vp]
    L.geot_sddmm_coo.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]
    L.geot_mh_sddmm_coo.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp]
    L.geot_gather_select_backward.argtypes = [c_vp] * 9 + [c_i64, c_i64, c_i64, c_i64, c_int, c_vp]
    L.geot_gather_rows.argtypes = [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]
    L.geot_csr_workspace_bytes.restype = c_sz
    L.geot_csr_workspace_bytes.argtypes = [c_i64, c_i64, c_i64, c_int]
    L.geot_csr_gws.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_coo_to_csr.argtypes = [c_vp, c_i64, c_i64, c_vp, c_int, c_vp]
    L.geot_slab_units.restype = c_int
    L.geot_slab_rows_per_group.argtypes = [c_int, c_i64]
    L.geot_slab_rows_per_group.restype = c_int
    L.geot_slab_rows_per_group_dtype.argtypes = [c_int, c_i64, c_int]
    L.geot_slab_rows_per_group_dtype.restype = c_int
    L.geot_slab_units_for.argtypes = [c_int, c_i64]
    L.geot_slab_units_for.restype = c_int
    L.geot_slab_rows_per_group_shape.argtypes = [c_int, c_i64, c_int, c_i64]
    L.geot_slab_rows_per_group_shape.restype = c_int
    L.geot_slab_workspace_bytes.argtypes = [ctypes.POINTER(SlabPlan), c_i64]
    L.geot_slab_workspace_bytes.restype = c_sz
    L.geot_slab_workspace_bytes_staged.argtypes = [ctypes.POINTER(SlabPlan), c_i64, c_int, c_i64, c_int]
    L.geot_slab_workspace_bytes_staged.restype = c_sz
    L.geot_slab_spmm.argtypes = [ctypes.POINTER(SlabPlan), c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_sz, c_vp]
    L.geot_slab_sddmm.argtypes = [ctypes.POINTER(SlabPlan), c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_slab_sddmm_staged.argtypes = [ctypes.POINTER(SlabPlan), c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_slab_to_plan_order.argtypes = [ctypes.POINTER(SlabPlan), c_vp, c_vp, c_i64, c_int, c_vp]
    L.geot_slab_mh_sddmm.argtypes = [ctypes.POINTER(SlabPlan), c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_sz, c_vp]
    L.geot_profile_enable.argtypes = [c_int]
    L.geot_profile_enable.restype = None
    L.geot_profile_reset.restype = None
    L.geot_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3 + [ctypes.POINTER(c_i64)]
    L.geot_profile_box.argtypes = [c_vp, c_sz, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_vp]
    L.geot_profile_box_rows.argtypes = [c_vp, c_i64, c_i64, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_vp, c_i64, c_int,
                                        ctypes.POINTER(ctypes.c_double), c_vp]
    L.geot_tune.argtypes = [c_int, c_int, c_int, c_int]
    L.geot_tune.restype = None
    L.geot_set_option.argtypes = [ctypes.c_char_p, c_int]
    L.geot_set_option.restype = c_int
    if L.geot_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {L.geot_abi_version()} != {ABI_VERSION}")
    _lib = L
    return L


def last_error() -> str:
    return load().geot_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != GEOT_OK:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


if __name__ == "__main__":      # `python3 geot_amd/_lib.py [lib|plugin|stamp]` (the Makefile's recipes)
    import sys
    what = sys.argv[1] if len(sys.argv) > 1 else "stamp"
    if what == "lib":
        build(verbose=True)
    elif what == "devlib":
        build(verbose=True, dev=True)
    elif what == "plugin":
        build(verbose=True)
        build_plugin(verbose=True)
    elif what == "stamp-plugin":
        stamp(PLUGIN_PATH)
    else:
        stamp()
