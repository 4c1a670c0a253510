#!/usr/bin/env python3
"""Instruction mix of one kernel's hot region from hipcc's assembly (no GPU needed): how many MFMA / VALU / SALU / LDS / VMEM / wait /
branch instructions sit between its first and last MFMA (+- a margin), the most frequent opcodes, scratch accesses and scalar spills
(v_writelane / v_readlane), and -- with --waits -- the s_waitcnt in front of each block of MFMAs.  Round 4's late gains came from reading
exactly this (a tile decode and a transform of zeros in the K loop's tail, loads at their point of use in epilogues).

usage: tools/debug/isa_mix.py <source without .hip, e.g. conv3x3_winox> <substring of the demangled kernel name> [--margin N] [--waits]
       EXTRA_HIPCC_FLAGS=... is honoured; conv3x3_winox is compiled with -fno-slp-vectorize like the product build."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, pat = sys.argv[1], sys.argv[2]
margin = int(sys.argv[sys.argv.index("--margin") + 1]) if "--margin" in sys.argv else 200
out = tempfile.mkdtemp()
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I%s/include" % ROOT, "-I%s/bsvd_amd/csrc" % ROOT, "-w", "-save-temps=obj"]
if src == "conv3x3_winox":
    flags.append("-fno-slp-vectorize")
flags += os.environ.get("EXTRA_HIPCC_FLAGS", "").split()
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", "%s/bsvd_amd/csrc/%s.hip" % (ROOT, src), "-o", "%s/%s.o" % (out, src)], check=True)
asm = open("%s/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % (out, src)).read()
names = sorted(set(re.findall(r"^(_Z\w+):", asm, re.M)))
if not names:
    sys.exit("no kernel symbols found")
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
for n, d in zip(names, dem):
    if pat not in d:
        continue
    i = asm.index("\n" + n + ":")
    lines = asm[i:asm.index("s_endpgm", i)].split("\n")
    mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
    print(d[:160])
    print("  %d lines, %d MFMAs (lines %s..%s); scratch stores %d loads %d; v_writelane %d v_readlane %d" % (
        len(lines), len(mf), mf[0] if mf else "-", mf[-1] if mf else "-", sum("scratch_store" in l for l in lines),
        sum("scratch_load" in l for l in lines), sum("v_writelane" in l for l in lines), sum("v_readlane" in l for l in lines)))
    if not mf:
        continue
    cls, ops = collections.Counter(), collections.Counter()
    for l in lines[max(0, mf[0] - margin):mf[-1] + margin]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        c = ("MFMA" if op.startswith("v_mfma") else "VALU" if op.startswith("v_") else "WAIT" if op.startswith("s_waitcnt") else
             "NOP" if op.startswith("s_nop") else "BRANCH" if "branch" in op else "BARRIER" if op.startswith("s_barrier") else
             "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("buffer_", "global_", "scratch_")) else "OTHER")
        cls[c] += 1
        ops[op] += 1
    print("  region mix:", dict(cls))
    print("  top opcodes:", ops.most_common(25))
    if "--waits" in sys.argv:
        prev = -10
        for k in mf:
            if k - prev > 40:
                w = [l.strip() for l in lines[k - 12:k + 1] if "s_waitcnt" in l]
                print("  MFMA block at line %d: waits in front %s" % (k, w))
            prev = k
