"""Windows of 200 instructions with >= 20 branch instructions in one kernel of an llvm-objdump listing: where wave-uniform
tests were left inside unrolled per-value code (every value its own chain of basic blocks).  Usage: listing kernel-name-part"""
import sys
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if name in l and l.endswith(">:")][0]
ends = [i for i in range(start + 1, len(lines)) if lines[i].endswith(">:")]
body = [l.strip().split("//")[0].rstrip() for l in lines[start + 1:(ends[0] if ends else len(lines))] if l.strip()]
hot = []
for i in range(0, len(body), 200):
    seg = body[i:i + 200]
    br = sum(1 for x in seg if x.startswith("s_cbranch"))
    if br >= 20:
        hot.append((i, br, sum(1 for x in seg if x.startswith("ds_")), sum(1 for x in seg if x.startswith("v_mfma")),
                    sum(1 for x in seg if x.startswith(("v_exp", "v_rcp", "v_rsq")))))
print(name, len(body), "instructions; windows (offset, branches, ds, mfma, transcendental):", hot)
