"""Opcode histogram of an instruction range of one function (indices as printed by asm_loops.py).
usage: python tools/diag/asm_hist.py file.s <function-substring> <first> <last>"""
import re, sys, collections
path, key, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_]\w*:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
insts = []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.?LBB", t):
        continue
    insts.append(t.split(";")[0].strip())
c = collections.Counter(t.split()[0] for t in insts[a:b + 1])
print(sum(c.values()), ", ".join("%s %d" % kv for kv in c.most_common(int(sys.argv[5]) if len(sys.argv) > 5 else 45)))
