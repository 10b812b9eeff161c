#!/usr/bin/env python3
"""Instruction mix of one kernel in a device assembly listing (hipcc -S --cuda-device-only): VALU / SALU / LDS / VMEM / scratch counts and
the v_readlane / v_writelane share (SGPR spills).   python profiles/tools/isa_count.py mapper.s '<mangled-name substring>'"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
txt = open(path).read()
# kernel bodies: "<name>:" ... ".Lfunc_end"
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.strip()]
    def n(pred): return sum(1 for i in ins if pred(i))
    valu = n(lambda i: i.startswith("v_"))
    print(name[:90])
    print("  total %d | VALU %d (v_readlane/v_writelane %d, v_cndmask %d, v_cmp %d) | SALU %d | LDS %d | VMEM %d | scratch %d | s_waitcnt %d" % (
        len(ins), valu, n(lambda i: i.startswith(("v_readlane", "v_writelane"))), n(lambda i: i.startswith("v_cndmask")), n(lambda i: i.startswith("v_cmp")),
        n(lambda i: i.startswith("s_") and not i.startswith("s_waitcnt")), n(lambda i: i.startswith("ds_")), n(lambda i: i.startswith(("global_", "buffer_", "flat_"))),
        n(lambda i: i.startswith("scratch_")), n(lambda i: i.startswith("s_waitcnt"))))
