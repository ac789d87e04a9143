"""The token level of the GPU DistEst / A-statistic parser
(gt-scaffold_amd/csrc/gts_deparse_tok.hpp, the source the device compiles)
compiled for the host and compared with the libc calls the reference makes --
sscanf("%[^>,],%ld,%ld,%f") (ref gt_scaffolder_parser.c:212, :340) and
sscanf("%f") (ref gt_scaffolder_algorithms.c:126) -- on random tokens: a token
the parser accepts must scan to exactly its values (floats bit for bit), a
token it rejects must not scan, a token it hands to the host may be anything
(tests/hostsim/deparse_fuzz.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tokens_against_sscanf(seed):
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.run(["make", "-s", "-C", d, "deparse_fuzz"], check=True)
    r = subprocess.run([os.path.join(d, "deparse_fuzz"), str(seed), "1000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    checked, records, fails, irregular, mismatches = map(int, r.stdout.split())
    assert mismatches == 0
    # the generator mostly writes regular tokens: most of them are decided here, not by the host
    assert checked > 1_500_000 and records > checked // 2 and fails > 100_000 and irregular < checked // 3
