#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a built library (the .amdhsa notes of its gfx950 code objects).

    python tools/diag/kernel_resources.py staticfusion_amd/csrc/libsf_hip.so [name-filter]
"""
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def elves(so_path):
    blob = open(so_path, "rb").read()
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            return
        count = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tlen].decode()
            off += tlen
            if "gfx950" in triple and size:
                yield blob[i + o:i + o + size]
        pos = i + 24


def main():
    so, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    for elf in elves(so):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(elf)
            f.flush()
            notes = subprocess.check_output([READELF, "--notes", f.name]).decode()
        for blk in notes.split("- .agpr_count:")[1:]:
            get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = get("name")
            if flt in name:
                print("%-60s vgpr %3s agpr %3s sgpr %3s  spills v %3s s %3s  scratch %5s B  lds %6s B" % (
                    name[:60], get("vgpr_count"), blk.split()[0], get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                    get("private_segment_fixed_size"), get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
