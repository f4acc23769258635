#!/usr/bin/env python3
"""Per-kernel resource usage of a built libgeot_hip.so (VGPRs, SGPRs, scratch, LDS) read from the gfx950 code objects
bundled in it - no GPU needed.  `python tools/kernel_resources.py [lib.so] [name-filter]`."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx" in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + 1


def kernels(path):
    out = []
    for triple, co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        finally:
            os.unlink(f.name)
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            out.append({"name": name, "vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "scratch": get("private_segment_fixed_size"),
                        "lds": get("group_segment_fixed_size"), "vgpr_spill": get("vgpr_spill_count")})
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "geot_amd", "libgeot_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'vgpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7} {'spill':>6}  kernel")
    for k in sorted(kernels(lib), key=lambda k: k["name"]):
        if flt in k["name"]:
            print(f"{k['vgpr']:>5} {k['sgpr']:>5} {k['scratch']:>8} {k['lds']:>7} {k['vgpr_spill']:>6}  {k['name'][:150]}")
