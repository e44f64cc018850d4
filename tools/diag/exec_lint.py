"""A lint over the device assembly for ONE miscompilation this repository has met (profiles/HISTORY.md, round 5: the "address 0" fault).

A divergent loop leaves through `s_andn2_b64 exec, exec, mask; s_cbranch_execz .Lexit`: the exit block is entered with EXEC = 0 and
re-enables its lanes with `s_or_b64 exec, exec, saved`. Anything the compiler places in front of that restore does nothing: VALU, LDS
and memory instructions are predicated on EXEC. LLVM keeps its own copies behind the restore ("block prologue") -- except, as seen
with ROCm 7.2's hipcc, when the source has a barrier there: `for (i = tid; i < n; i += NT) {...} __syncthreads();` INSIDE a branch the
compiler takes for divergent (its condition came out of a VGPR although every lane holds the same value). The exit block then reads
    v_mov_b64 v[62:63], v[92:93]      <- a live-range split copy, executed with EXEC = 0: the stream-state pointer is NOT restored
    s_barrier
    s_or_b64 exec, exec, s[10:11]
and the next use of v[62:63] stores through whatever the loop left there.

The lint: for every label that a `s_cbranch_execz` targets, the instructions from the label to the first write of EXEC; flagged when
an `s_barrier` sits before that write together with any vector instruction in front of the barrier. A barrier is a point every lane
of the workgroup passes, so EXEC = 0 there can only be a structurizer artefact.
usage: exec_lint.py file.s | lib.so [...]   (exit code 1 if flagged). A shared library is taken apart: every gfx950 code object of its
offload bundles is disassembled with llvm-objdump (tests/test_capi_and_host.py runs this over the built libraries).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LABEL = r"(?:\.LBB\d+_\d+|L\d+)"


def code_objects(so_path):
    """The gfx950 code objects inside the clang offload bundles of a HIP shared library, as disassembly text (one per object)."""
    blob = open(so_path, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            break
        count = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tlen].decode()
            off += tlen
            if "gfx950" in triple and size:
                with tempfile.NamedTemporaryFile(suffix=".elf") as f:
                    f.write(blob[i + o:i + o + size])
                    f.flush()
                    out.append(subprocess.check_output([OBJDUMP, "-d", "--symbolize-operands", f.name]).decode())
        pos = i + 24
    return out


def lint(path):
    if path.endswith(".so"):
        flagged = []
        for text in code_objects(path):
            flagged += lint_text(text)
        return flagged
    return lint_text(open(path).read())


def lint_text(text):
    lines = [re.sub(r"\s*//.*$", "", l) for l in text.splitlines()]  # (objdump appends the encoding as a // comment)
    lines = [re.sub(r"^(?:[0-9a-f]+ )?<(L\d+)>:", r"\1:", l) for l in lines]  # objdump: "000000000002fe74 <L1127>:"
    targets = set()
    for l in lines:
        m = re.match(r"\s+s_cbranch_execz\s+(" + LABEL + r")", l)
        if m:
            targets.add(m.group(1))
    flagged, func = [], None
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(?:[0-9a-f]+ <)?([A-Za-z_]\w*)>?:", l)
        if m and not l.startswith(".L") and not re.match(r"^L\d+:", l):
            func = m.group(1)
        m = re.match(r"^(" + LABEL + r"):", l)
        if m and m.group(1) in targets:
            vec, j, barrier_at = [], i + 1, None
            while j < len(lines):
                t = lines[j].split(";")[0].strip()
                j += 1
                if not t or t.startswith(".") or re.match(r"^" + LABEL + r":", t):
                    continue  # (a label in between: falls through, the same straight line)
                op = t.split()[0]
                if re.search(r"\bexec\b", t) and (op.startswith("s_or") or op.startswith("s_mov") or op.startswith("s_and") or op.startswith("s_xor") or op.startswith("s_andn2")) and re.match(r"s_\w+\s+exec", t):
                    break  # EXEC written: whatever follows runs with the restored mask
                if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_swappc")):
                    break
                if op == "s_barrier":
                    barrier_at = j
                    continue
                if op.startswith(("v_", "ds_", "global_", "flat_", "buffer_", "scratch_")) and barrier_at is None:
                    if op in ("v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32"):
                        continue  # lane-indexed: do not depend on EXEC
                    vec.append(t)
            if barrier_at is not None and vec:
                flagged.append((func, m.group(1), vec))
        i += 1
    return flagged


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for func, label, vec in lint(p):
            bad += 1
            print("%s: %s %s: executed with EXEC = 0 in front of a barrier: %s" % (p, func, label, "; ".join(vec[:4])))
    print("exec_lint: %d block(s) flagged in %d file(s)" % (bad, len(sys.argv) - 1))
    sys.exit(1 if bad else 0)
