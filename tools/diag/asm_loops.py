"""Static view of the device assembly of one function: its loops (backward branches) with instruction counts by class.
usage: python tools/diag/asm_loops.py /tmp/frame_nt256.s <function-substring> [min_insts]
(the assembly: hipcc <flags of the Makefile> -DSF_NT=256 -S --cuda-device-only -o /tmp/frame_nt256.s sf_frame_kernels.hip)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_]\w*:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
insts = []  # (index, text)
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    insts.append(t.split(";")[0].strip())
def cls(t):
    op = t.split()[0]
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "scratch" if op.startswith("scratch_") else "vmem"
    return "other"
total = {}
for t in insts:
    total[cls(t)] = total.get(cls(t), 0) + 1
print("function %s: %d instructions %s" % (key, len(insts), total))
loops = []
for i, t in enumerate(insts):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt <= i:
            loops.append((tgt, i))
for a, b in sorted(set(loops)):
    if b - a + 1 < min_insts:
        continue
    c = {}
    for t in insts[a:b + 1]:
        c[cls(t)] = c.get(cls(t), 0) + 1
    inner = [x for x in loops if x[0] >= a and x[1] <= b and x != (a, b)]
    print("  loop insts %5d..%5d (%4d)%s  %s" % (a, b, b - a + 1, "  [has %d inner]" % len(set(inner)) if inner else "", c))
