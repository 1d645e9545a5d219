"""Per-kernel resource table from `make EXTRA=-Rpass-analysis=kernel-resource-usage 2> log`:  python tools/isa/resource_report.py log"""
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0][:100]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-100s VGPR %3s AGPR %3s spillV %3s spillS %3s scratch %4s occ %s" % (
        name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
