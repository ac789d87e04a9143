"""The component kernels hold no scratch and no spilled vector registers, and the
dominant one stays under the register count that the launch geometry needs
(one build of k_components_pool lost vertex states next to 52
spilled VGPRs and 164 B of scratch per lane; the cause of the spills -- lane-derived
values hoisted out of the kernel's component loop -- is gone since GtsWave64::lane()
is opaque, and this test keeps it that way).  Reads the code object's metadata, no GPU."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.fixture(scope="module")
def res(tmp_path_factory):
    from kernel_resources import resources
    return resources(str(tmp_path_factory.mktemp("isa") / "gts_engine_dev.s"))


def kernels(res, pattern):
    ks = {k: v for k, v in res.items() if re.search(pattern, k)}
    assert ks, pattern
    return ks


def test_component_kernels_have_no_scratch(res):
    for name, r in kernels(res, r"k_components(_lds|_team)?11GtsCompView|k_components_(pool|fast)PK11GtsCompView|k_components_fast2ILi10E|"
                                r"k_walk_tasks|k_select_walks").items():
        assert r["vspill"] == 0 and r["scratch"] == 0, (name, r)


def test_dominant_kernel_register_budget(res):
    # one workgroup of 16 wavefronts per CU needs <= 128; the lean program is held to 96
    # (five wavefronts per SIMD) so that the geometry can change without a rewrite
    for name, r in kernels(res, r"k_components_fastPK11GtsCompView").items():
        assert r["vgpr"] <= 96, (name, r)
    for name, r in kernels(res, r"k_components_fast2ILi10E").items():
        assert r["vgpr"] <= 96, (name, r)
    for name, r in kernels(res, r"k_components_poolPK11GtsCompView").items():
        assert r["vgpr"] <= 128, (name, r)


def test_write_back_loops_are_not_unrolled():
    src = open(os.path.join(ROOT, "gt-scaffold_amd", "csrc", "gts_component.hpp")).read()
    # run(): the loop that writes the vertex states back
    m = re.search(r"#pragma unroll 1\s*\n\s*for \(uint32_t s = lane; s < nv; s \+= W::WIDTH\) \{\s*\n\s*const uint8_t st = M\.vst\[s\];", src)
    assert m, "run(): the write-back loop lost its '#pragma unroll 1'"
    # run_fast(): both write-back loops
    assert len(re.findall(r"#pragma unroll 1\s*\n\s*for \(uint32_t (s|k) = lane;", src)) >= 3


def test_build_and_filter_kernels_keep_their_occupancy(res):
    """what round 4's measurements rest on: the streaming kernels of the build stage hold no
    scratch, the radix pass stays at four wavefronts a SIMD (two 512-thread workgroups a CU:
    forcing more made it spill, 1.9 -> 11.6 ms), and the pair kernels of the filter stage
    3072 edges a workgroup so that three of them share a CU's LDS (3.24 -> 2.44 ms)"""
    for name, r in kernels(res, r"k_pair_segments_bucket|k_gather_csrILi4|k_twinsILi4|k_emit_edges|k_radix_scatter").items():
        assert r["vspill"] == 0 and r["scratch"] == 0, (name, r)
    for name, r in kernels(res, r"k_radix_scatter").items():
        assert r["vgpr"] <= 128, (name, r)
    src = open(os.path.join(ROOT, "gt-scaffold_amd", "csrc", "gts_engine.hip")).read()
    m = re.search(r"#define GTS_FP_CAP (\d+)", src)
    assert m and 17 * int(m.group(1)) * 3 <= 160 * 1024, "three workgroups of k_filter_pairs no longer fit a CU's LDS"

