#!/usr/bin/env python3
"""Static instruction counts of the loops of one kernel (offline, no GPU):

    python scripts/isa_loops.py polytope_amd/csrc/plp_reduce_r.hip 'reduce_r_kernelILi3ELi4' [extra hipcc flags]

Compiles the file to gfx950 assembly, cuts out the kernel whose mangled name contains the pattern,
prints VGPR / scratch / occupancy from the kernel descriptor and, for every backward branch (loop),
the number of VALU / SALU / DS / VMEM instructions in its body.  Used to judge an edit of the
pivot loop before spending GPU time on it.
"""
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-S", "--cuda-device-only", "-o", f.name, src] + flags, stderr=subprocess.DEVNULL)
    text = open(f.name).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))  # blocks may sit after s_endpgm
body = text[start:end + 1]
name = text[start].rstrip(":")
for key in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
    for l in text[end:end + 120]:
        m = re.match(r";\s+%s:\s+(\d+)" % key, l)
        if m:
            print("%-12s %s" % (key, m.group(1)))
            break
kind = lambda l: ("valu" if re.match(r"\s+v_", l) else "salu" if re.match(r"\s+s_", l) else
                  "ds" if re.match(r"\s+ds_", l) else "vmem" if re.match(r"\s+(global|buffer|scratch|flat)_", l) else None)
tot = {}
for l in body:
    k = kind(l)
    if k:
        tot[k] = tot.get(k, 0) + 1
print("whole kernel:", tot)
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", l)] if m}
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        s = labels[m.group(1)]
        c = {}
        for x in body[s:i + 1]:
            k = kind(x)
            if k:
                c[k] = c.get(k, 0) + 1
        if c.get("valu", 0) >= 100:
            print("loop %-10s lines %5d-%5d  %s" % (m.group(1), s, i, c))
