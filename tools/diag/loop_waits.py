#!/usr/bin/env python3
"""The innermost loops of one device function in a BUILT library (or object): instruction counts by class and the vmcnt of every
wait in them -- whether the loads a loop issues ahead stay in flight (vmcnt(N), N > 0) or every wait is a full one.

    python tools/diag/loop_waits.py staticfusion_amd/csrc/libsf_hip.so 256o5 solve_linearise_stripsILb0ELb0
                                     (library or .o)                   (build: 256 | 256o5 | 1024 | cluster)  (mangled-name part)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exec_lint  # noqa: E402


def innermost_loops(lib, kern, fn, verbose=False):
    """[{valu, loads, stores, flat, scratch, dpp, lds, waits: [vmcnt, ...]}, ...] of function `fn` in build `kern` of `lib`"""
    out = []
    for text in exec_lint.code_objects(lib):
        kernels = re.findall(r"<(_Z\d+sf_frame_kernel\w*)>:", text)
        if not any(k.endswith("nt" + kern + "PK5KArgs11FrameLaunch") for k in kernels):
            continue
        m = re.search(r"^[0-9a-f]+ <_Z\d+%s[^>]*>:\n" % re.escape(fn), text, flags=re.M)
        if not m:
            continue
        rest = text[m.end():]
        e = re.search(r"^[0-9a-f]+ <_Z[^>]+>:\n", rest, flags=re.M)
        lines = (rest[:e.start()] if e else rest).splitlines()
        if verbose:
            print("%s in sf_frame_kernel_nt%s: %d lines, %d scratch accesses" % (fn, kern, len(lines), sum("scratch_" in l for l in lines)))
        label = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^[0-9a-f]+ <(L\d+)>:", l)
            if mm:
                label[mm.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"\b(s_cbranch\w*|s_branch)\s+(L\d+)", l)
            if mm and mm.group(2) in label and label[mm.group(2)] < i:
                loops.append((label[mm.group(2)], i))
        for a, b in loops:
            if any((c, d) != (a, b) and c >= a and d <= b for c, d in loops):
                continue  # not innermost
            seg = lines[a:b]
            count = lambda pred: sum(1 for x in seg if pred(x))
            rec = {"lines": (a, b), "valu": count(lambda x: re.match(r"\s*v_", x) is not None), "loads": count(lambda x: "global_load" in x),
                   "stores": count(lambda x: "global_store" in x), "flat": count(lambda x: "flat_" in x), "scratch": count(lambda x: "scratch_" in x),
                   "dpp": count(lambda x: "dpp" in x), "lds": count(lambda x: re.match(r"\s*ds_", x) is not None),
                   "waits": [int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x in seg if "vmcnt" in x]}
            out.append(rec)
            if verbose:
                print("  loop at lines %d..%d: VALU %d, global loads %d, global stores %d, flat %d, scratch %d, DPP %d, LDS %d; vmcnt of its waits: %s" % (
                    a, b, rec["valu"], rec["loads"], rec["stores"], rec["flat"], rec["scratch"], rec["dpp"], rec["lds"], " ".join(map(str, rec["waits"])) or "-"))
    return out


def main():
    innermost_loops(sys.argv[1], sys.argv[2], sys.argv[3], verbose=True)

if __name__ == "__main__":
    main()
