"""Registers, spills and LDS of the engine's kernels, from the device assembly.
usage: python tools/kernel_resources.py [regex]   (compiles gts_engine.hip with -S into /tmp)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gt-scaffold_amd", "csrc")
out = "/tmp/gts_engine_dev.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-w",
                "-I" + os.path.join(root, "include"), "-I" + src, "--offload-device-only", "-S",
                os.path.join(src, "gts_engine.hip"), "-o", out], check=True)
t = open(out).read()
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
for b in t.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if not pat.search(name):
        continue
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    print(name, "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "vspill", g("vgpr_spill_count"),
          "sspill", g("sgpr_spill_count"), "scratch", g("private_segment_fixed_size"))
