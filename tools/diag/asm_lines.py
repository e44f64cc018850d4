"""VALU instructions of one device function by SOURCE LINE (static counts): compile with -gline-tables-only -S and attribute every
instruction to the last `.loc` in front of it -- which source lines of a stage the compiler spent its vector instructions on.
Inlined code is attributed to the line of the inlined body (the innermost location).
usage: python tools/diag/asm_lines.py file.s <function-substring> [file-number-filter] [top]
(the assembly: hipcc <Makefile flags> -DSF_NT=256 -DSF_OCC=5 -gline-tables-only -S --cuda-device-only -o file.s sf_frame_kernels.hip)"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
lines = open(path).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_]\w*:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = (0, 0)
valu = collections.Counter()
alln = collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.?LBB", t):
        continue
    op = t.split()[0]
    alln[cur] += 1
    if op.startswith("v_"):
        valu[cur] += 1
print("function %s: %d VALU of %d instructions" % (key, sum(valu.values()), sum(alln.values())))
byfile = collections.Counter()
for (f, ln), n in valu.items():
    byfile[f] += n
print("  by file:", ", ".join("%s %d" % (files.get(f, f), n) for f, n in byfile.most_common()))
items = [(k, n) for k, n in valu.items() if only is None or k[0] == only]
for (f, ln), n in sorted(items, key=lambda kv: -kv[1])[:top]:
    print("  %-24s line %5d: %4d VALU (%4d all)" % (files.get(f, f), ln, n, alln[(f, ln)]))
