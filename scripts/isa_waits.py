#!/usr/bin/env python3
"""Build host only: the s_waitcnt vmcnt(N) histogram, register count, scratch and occupancy of every kernel of one source file.

    python scripts/isa_waits.py [grb_mxv.hip] [name filter]

A software-pipelined loop whose every wait is vmcnt(0) is not pipelined: the compiler derives N from the number of memory instructions
between a request and its use and must take the path with the fewest -- loads under a condition count as absent (DESIGN.md section
4.1.0, "the pipeline that was not one": k_mxv_hstrip lost 55 of its 207 us to it for a round).  Run this after touching a pipelined
kernel; the shipped k_mxv_hstrip kernels show vmcnt(13..21) for most of their waits."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "python-graphblas_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 else "grb_mxv.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-Wno-unused-result", "-munsafe-fp-atomics", "--cuda-device-only", "-S", os.path.join(SRC, src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
rows = []
for n, (i, name) in enumerate(starts):
    body = lines[i: starts[n + 1][0] if n + 1 < len(starts) else len(lines)]
    text = "\n".join(body)
    if ".amdhsa_kernel" not in text and "s_endpgm" not in text:
        continue
    waits = collections.Counter(int(x) for x in re.findall(r"vmcnt\((\d+)\)", text))
    meta = {k: (re.search(r"; %s: (\d+)" % k, text) or [None, "?"])[1] for k in ("NumVgprs", "ScratchSize", "Occupancy")}
    rows.append((name, waits, meta))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, waits, meta), dn in zip(rows, names):
    dn = dn.split("(")[0]
    if flt and flt not in dn:
        continue
    total, zero = sum(waits.values()), waits.get(0, 0)
    top = ", ".join("vmcnt(%d) x %d" % kv for kv in sorted(waits.items(), key=lambda kv: -kv[1])[:5])
    print("%-70s vgpr %3s scratch %4s occ %s | waits %3d, vmcnt(0) %3d | %s" % (dn[:70], meta["NumVgprs"], meta["ScratchSize"], meta["Occupancy"], total, zero, top))
