#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects inside a built library.

    python tools/kernel_resources.py [gpismap_amd/libgpismap_amd.so] [name-filter]

Reads the .hip_fatbin section (clang offload bundles, one per translation unit), takes every amdgcn code object and
parses `llvm-readelf --notes` (the AMDGPU metadata: .vgpr_count, .vgpr_spill_count, .private_segment_fixed_size, ...).
tests/test_host.py uses kernel_table() to keep the ongpis_* kernels free of spills and scratch."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "x.o")])
        data = open(fat, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size > 0:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def kernel_table(lib):
    """-> {kernel name: {vgpr, agpr, sgpr, spill, scratch, lds, triple}}"""
    table = {}
    for triple, blob in _code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and "name" in cur:      # first field of a new kernel record
                cur = {}
            cur[k] = v
            if k == "wavefront_size" and "name" in cur:   # last field of a kernel record
                table[cur["name"]] = dict(vgpr=int(cur.get("vgpr_count", 0)), agpr=int(cur.get("agpr_count", 0)),
                                          sgpr=int(cur.get("sgpr_count", 0)), spill=int(cur.get("vgpr_spill_count", 0)),
                                          sgpr_spill=int(cur.get("sgpr_spill_count", 0)),
                                          scratch=int(cur.get("private_segment_fixed_size", 0)),
                                          lds=int(cur.get("group_segment_fixed_size", 0)), triple=triple)
                cur = {}
    return table


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpismap_amd", "libgpismap_amd.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    t = kernel_table(lib)
    dm = demangle(list(t))
    print("%-6s %-5s %-5s %-6s %-8s %-7s kernel" % ("vgpr", "agpr", "sgpr", "spill", "scratch", "lds"))
    for k in sorted(t, key=lambda n: dm[n]):
        if flt and flt not in dm[k]:
            continue
        r = t[k]
        short = re.sub(r"\(.*", "", dm[k]).replace("void gpis::", "")
        print("%-6d %-5d %-5d %-6d %-8d %-7d %s" % (r["vgpr"], r["agpr"], r["sgpr"], r["spill"], r["scratch"], r["lds"], short))
