#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels inside libdeme_hip.so (or a variant): pulls the gfx950 code object out of the
HIP fat binary and reads its msgpack notes with llvm-readelf.   usage: co_info.py lib.so [kernel-name-substring ...]"""
import re
import struct
import subprocess
import sys
import tempfile

path, pats = sys.argv[1], sys.argv[2:]
blob = open(path, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
elfs = []  # one code object per translation unit linked into the library
at = blob.find(magic)
if at < 0:
    sys.exit("no offload bundle in " + path)
while at >= 0:
    n = struct.unpack_from("<Q", blob, at + 24)[0]
    pos = at + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24:pos + 24 + tl].decode()
        pos += 24 + tl
        if "gfx950" in triple:
            elfs.append(blob[at + off:at + off + size])
    at = blob.find(magic, at + 1)
if not elfs:
    sys.exit("no gfx950 code object")
txt = ""
for elf in elfs:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf), f.flush()
        txt += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or (pats and not any(p in name.group(1) for p in pats)):
        continue
    get = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(f"{name.group(1)[:90]:90s} vgpr {get('vgpr_count'):>3s} sgpr {get('sgpr_count'):>3s} lds {get('group_segment_fixed_size'):>6s} "
          f"scratch {get('private_segment_fixed_size'):>4s} vspill {get('vgpr_spill_count')} sspill {get('sgpr_spill_count')}")
