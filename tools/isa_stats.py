#!/usr/bin/env python3
"""ISA statistics of one kernel of a gfx950 build (no GPU needed): instruction classes, scratch traffic and where it sits.
usage: tools/isa_stats.py pgq_meet.hip k_meet4dILb0ELb0 [extra hipcc flags]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(root, "duckpgq-extension_amd", "csrc")
asm = "/tmp/isa_%s.s" % os.path.basename(src)
if not os.environ.get("ISA_REUSE") or not os.path.exists(asm):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + root + "/include",
                           "-I" + csrc, "-DNDEBUG", "-S", "--cuda-device-only", "-o", asm, os.path.join(csrc, src)] + sys.argv[3:],
                          stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN3pgq.*:", l) and pat in l]
for st in start:
    end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[st:end]
    ins = [l.strip() for l in body if re.match(r"\s+[a-z_0-9]+\s", l) and not l.strip().startswith(";") and not l.strip().startswith(".")]
    cls = {"valu": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0, "scratch": 0, "exec": 0, "readlane": 0, "writelane": 0, "waitcnt": 0, "branch": 0}
    for l in ins:
        op = l.split()[0]
        if op.startswith("scratch_"): cls["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): cls["vmem"] += 1
        elif op.startswith("ds_"): cls["lds"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): cls["smem"] += 1
        elif op.startswith("s_waitcnt"): cls["waitcnt"] += 1
        elif op.startswith(("s_cbranch", "s_branch")): cls["branch"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"): cls["readlane"] += 1
        elif op.startswith("v_writelane"): cls["writelane"] += 1
        elif op.startswith("v_"): cls["valu"] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
            if "exec" in l: cls["exec"] += 1
    print(lines[st].split(":")[0][:70])
    print("  instructions %d: %s" % (len(ins), cls))
    # scratch instructions with the nearest preceding label (loop context)
    lab = ""
    for l in body:
        if re.match(r"^\.LBB", l): lab = l.split(":")[0]
        if "scratch_" in l: print("   ", lab, l.strip()[:90])
