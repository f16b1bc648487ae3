#!/usr/bin/env python3
"""Compact view of a kernel's instruction schedule from `hipcc -S` output (build container only, no GPU).

usage: sched.py file.s '<demangled-name substring>' [--loop]      (c++filt must be on PATH)
One character per instruction: M mfma, r ds_read, w ds_write, g global/buffer load, S global/buffer store,
v VALU, s SALU, W s_waitcnt, B s_barrier, b branch, . other.  With --loop only blocks marked "in Loop" are printed.
"""
import re, subprocess, sys

def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if op.startswith(("buffer_load", "global_load", "flat_load")): return "g"
    if op.startswith(("buffer_store", "global_store", "flat_store", "global_atomic", "buffer_atomic")): return "S"
    if op == "s_waitcnt": return "W"
    if op == "s_barrier": return "B"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "b"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "s"
    return "."

def main():
    path, pat = sys.argv[1], sys.argv[2]
    loop_only = "--loop" in sys.argv
    text = open(path).read().split("\n")
    syms = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(s for _, s in syms), capture_output=True, text=True).stdout.split("\n")
    for (i, s), d in zip(syms, dem):
        if pat in d:
            print("==", d)
            j = i + 1
            cur, name, inloop = [], "entry", False
            def flush():
                if cur and (inloop or not loop_only):
                    st = "".join(cur)
                    print(f"{name:14s} {'L' if inloop else ' '} n={len(st):4d} mfma={st.count('M'):3d} {st}")
            while j < len(text) and not text[j].startswith(".Lfunc_end"):
                l = text[j].strip()
                if re.match(r"^\.LBB\d+_\d+:", l):
                    flush(); cur = []; name = l.split(":")[0]; inloop = "in Loop" in l or "Loop Header" in l
                elif l and not l.startswith((";", ".")):
                    cur.append(classify(l))
                    if l.startswith("s_waitcnt"):
                        pass
                j += 1
            flush()
            m = [l for l in text[j:j + 400] if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize", l)]
            print("\n".join(m[:6]))
main()
