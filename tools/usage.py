"""Dev tool: per-kernel register / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
usage: python tools/usage.py file.usage.txt [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    dem = re.sub(r"^void attwarp::", "", dem)
    print("%-100s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %6s" % (dem[:100], g("VGPRs"), g("AGPRs"), g("SGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
