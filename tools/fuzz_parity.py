"""Randomised parity sweep on the GPU box: small synthetic graphs with varied
generator parameters and engine options, every stage against the oracle
(test infrastructure: uses tests/ helpers).
usage: python tools/fuzz_parity.py [seconds] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import make_inputs, oracle_from_inputs, pkg  # noqa: E402
from test_gpu_parity import engine_from_inputs  # noqa: E402


def one(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 6000))
    kw = dict(p_chimeric=float(rng.choice([0.0, 0.02, 0.08, 0.2])),
              p_bubble=float(rng.choice([0.0, 0.05, 0.15])),
              p_repeat=float(rng.choice([0.0, 0.03])),
              p_inversion=float(rng.choice([0.0, 0.3, 1.0])),
              p_relist=float(rng.choice([0.0, 0.05])),
              links_per_side=int(rng.choice([1, 2, 4])))
    opts = {}
    if rng.random() < 0.3:
        opts["defer_min_contigs"] = int(rng.choice([0, 16, 64]))
    if rng.random() < 0.3:
        opts["defer_ref_min_contigs"] = int(rng.choice([0, 8, 48]))
    if rng.random() < 0.3:
        opts["defer_unclean_work"] = int(rng.choice([0, 16, 2048]))
    if rng.random() < 0.2:
        opts["pool_components"] = 0
    if rng.random() < 0.2:
        opts["lds_int16_distances"] = 0
    if rng.random() < 0.15:
        opts["small_masks"] = 0
    if rng.random() < 0.15:
        opts["batch_walks"] = 0
    if rng.random() < 0.1:
        opts["lds_components"] = 0
    g = make_inputs(n, 7000 + seed, **kw)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g, **opts)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(0.01, 1.5, 400); eng.filter(0.01, 1.5, 400)
    if seed % 3 == 0:
        og.removecycles(); eng.removecycles()
    og.makescaffold(True); eng.makescaffold()
    ok = (np.array_equal(eng.vertex_states(), og.vertex_states()) and
          np.array_equal(eng.edge_states(), og.edge_states()))
    st = (eng.stat("components"), eng.stat("deferred_components"), eng.stat("slow_walks"))
    eng.close()
    return ok, n, kw, opts, st


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    done = bad = 0
    while time.time() - t0 < budget:
        ok, n, kw, opts, st = one(seed)
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, n, kw, opts, st, flush=True)
        if done % 25 == 0:
            print("...", done, "graphs,", bad, "mismatches, %.0f s" % (time.time() - t0), flush=True)
        seed += 1
    print("fuzz_parity: %d graphs, %d mismatches (seeds up to %d)" % (done, bad, seed - 1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
