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


def params(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 6000))
    kw = dict(p_chimeric=float(rng.choice([0.0, 0.02, 0.08, 0.2])),
              p_bubble=float(rng.choice([0.0, 0.05, 0.15])),
              p_repeat=float(rng.choice([0.0, 0.03])),
              p_inversion=float(rng.choice([0.0, 0.3, 1.0])),
              p_relist=float(rng.choice([0.0, 0.05])),
              links_per_side=int(rng.choice([1, 2, 4])))
    if seed >= 5000:
        # second family: ties (few distinct distances, short contigs), flipped
        # re-listings, larger graphs
        if rng.random() < 0.5:
            kw["dist_range_small"] = True
            kw["contig_median"] = int(rng.choice([250, 900]))
        if rng.random() < 0.3:
            kw["p_relist_flip"] = 0.1
        if rng.random() < 0.3:
            kw["unique_pairs"] = True
        if rng.random() < 0.2:
            n = int(rng.integers(6000, 25000))
            kw["p_chimeric"] = min(kw["p_chimeric"], 0.02)
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
    # round 4 (drawn last: the earlier draws of a seed stay what they were): the ways the
    # records of a contig pair are brought together in the build
    if rng.random() < 0.2:
        opts["pair_bucket_limit"] = int(rng.choice([1, 4, 64]))
    if rng.random() < 0.1:
        opts["pair_sort_full"] = 1
    return n, kw, opts


def oracle_states(seed):
    """the oracle's states after every stage the engine is asked for"""
    n, kw, _ = params(seed)
    g = make_inputs(n, 7000 + seed, **kw)
    og = oracle_from_inputs(g)
    og.mark_repeats(); og.filter(0.01, 1.5, 400)
    if seed % 3 == 0:
        og.removecycles()
    og.makescaffold(True)
    return og.vertex_states(), og.edge_states()


def oracle_worker():
    """child process (started before the parent touches the GPU): the reference's
    label-correcting search can push a queue of any length on a graph with
    inversions and misjoins (the engine bounds it, max_walk_pops); under an
    address-space limit such a graph ends this process instead of the box"""
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (16 << 30, 16 << 30))
    for line in sys.stdin:
        seed, path = line.split()
        v, e = oracle_states(int(seed))
        np.savez(path, v=v, e=e)
        print("ok", flush=True)


class Oracle:
    """oracle workers, all started up front: a process that has initialised the
    GPU must not start another program (fork + exec) on this pool"""

    def __init__(self, spare=40):
        import subprocess
        self.ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-worker"],
                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
                   for _ in range(spare)]

    def get(self, seed, path):
        while self.ps and self.ps[0].poll() is not None:
            self.ps.pop(0)
        if not self.ps:
            raise SystemExit("fuzz_parity: no oracle worker left")
        p = self.ps[0]
        p.stdin.write("%d %s\n" % (seed, path)); p.stdin.flush()
        if p.stdout.readline().strip() != "ok":
            p.wait(); self.ps.pop(0)
            return None
        z = np.load(path)
        return z["v"], z["e"]

    def close(self):
        for p in self.ps:
            p.stdin.close()
        for p in self.ps:
            p.wait()


def one(seed, oracle, tmp):
    n, kw, opts = params(seed)
    want = oracle.get(seed, tmp)
    if want is None:
        return None, n, kw, opts, None
    g = make_inputs(n, 7000 + seed, **kw)
    eng = engine_from_inputs(g, **opts)
    eng.mark_repeats()
    eng.filter(0.01, 1.5, 400)
    if seed % 3 == 0:
        eng.removecycles()
    eng.makescaffold()
    ok = np.array_equal(eng.vertex_states(), want[0]) and np.array_equal(eng.edge_states(), want[1])
    st = (eng.stat("components"), eng.stat("deferred_components"), eng.stat("slow_walks"))
    eng.close()
    return ok, n, kw, opts, st


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--oracle-worker":
        return oracle_worker()
    import tempfile
    oracle = Oracle()
    tmp = os.path.join(tempfile.mkdtemp(), "want.npz")
    skipped = 0
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import psutil
    proc = psutil.Process()
    t0 = time.time()
    done = bad = 0
    rss0 = proc.memory_info().rss
    while time.time() - t0 < budget:
        ok, n, kw, opts, st = one(seed, oracle, tmp)
        if ok is None:
            skipped += 1
            print("skipped seed", seed, n, kw, "(the oracle ran out of its 16 GB)", flush=True)
            seed += 1
            continue
        done += 1
        rss1 = proc.memory_info().rss
        if rss1 - rss0 > (256 << 20):     # host memory of the process grew by more than 256 MB in one graph
            print("RSS +%d MB at seed %d: n %d %s %s %s" % ((rss1 - rss0) >> 20, seed, n, kw, opts, st), flush=True)
        rss0 = rss1
        if rss1 > (40 << 30):
            print("fuzz_parity: stopping, the process holds %d GB" % (rss1 >> 30), flush=True)
            break
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, n, kw, opts, st, flush=True)
        if done % 25 == 0:
            import torch
            free, total = torch.cuda.mem_get_info()
            try:
                cg = int(open("/sys/fs/cgroup/memory.current").read()) >> 20
            except OSError:
                cg = -1
            print("...", done, "graphs,", bad, "mismatches, %.0f s; device memory in use %d MB, cgroup memory %d MB, RSS %d MB"
                  % (time.time() - t0, (total - free) >> 20, cg, rss1 >> 20), flush=True)
            if (total - free) > (60 << 30) or cg > (60 << 10):
                print("fuzz_parity: stopping, memory keeps growing", flush=True)
                break
        seed += 1
    oracle.close()
    print("fuzz_parity: %d graphs, %d mismatches, %d skipped (seeds up to %d)" % (done, bad, skipped, seed - 1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
