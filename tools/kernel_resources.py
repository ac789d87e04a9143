"""Registers, spills and LDS of the engine's kernels, from the device assembly.
usage: python tools/kernel_resources.py [regex]   (compiles gts_engine.hip with -S into /tmp)"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gt-scaffold_amd", "csrc")


def resources(out="/tmp/gts_engine_dev.s"):
    """{mangled kernel name: dict(vgpr, sgpr, vspill, sspill, scratch)} of gts_engine.hip for gfx950."""
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-w",
                    "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "--offload-device-only", "-S",
                    os.path.join(SRC, "gts_engine.hip"), "-o", out], check=True)
    t = open(out).read()
    res = {}
    for b in t.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda k: int(re.search(r"\.%s:\s+(\S+)" % k, b).group(1))
        res[name] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                         sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"))
    return res


if __name__ == "__main__":
    pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
    for name, r in resources().items():
        if pat.search(name):
            print(name, "vgpr", r["vgpr"], "sgpr", r["sgpr"], "vspill", r["vspill"], "sspill", r["sspill"],
                  "scratch", r["scratch"])
