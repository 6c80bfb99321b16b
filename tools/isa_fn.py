"""Dev tool: instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: python tools/isa_fn.py file.s <substring of the mangled kernel name>"""
import re, sys, collections
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)\.Lfunc_end" % re.escape(pat), txt, re.S | re.M):
    body = m.group(2)
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter()
    for i in ins:
        k = ("valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else
             "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        c[k] += 1
    print(m.group(1)[:90], len(ins), dict(c))
