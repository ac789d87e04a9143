"""The N > 1 path on CPU: component-partition step (label / plan / route) with
world_size 2 over gloo (real torch.distributed processes) and over the
in-process communicator.  The local labelling kernel is replaced by a numpy
union-find test double; the collectives, the plan and the routing are the
product code (gt-scaffold_amd/dist.py)."""
import os
import threading

import numpy as np
import pytest
import torch

from helpers import make_inputs, pkg

dist_mod = pkg.dist
CUTS = dict(copy_num_cutoff=0.3, astat_cutoff=20.0, pcutoff=0.01, cncutoff=1.5, ocutoff=400)


def numpy_label_fn(labels, root, ctg, skip):
    """test double of gtsg_label_components: same contract"""
    parent = labels.numpy().copy()

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    sk = skip.numpy()
    for a, b in zip(root.numpy(), ctg.numpy()):
        if sk[a] or sk[b]:
            continue
        ra, rb = find(a), find(b)
        if ra != rb:
            if ra < rb:
                parent[rb] = ra
            else:
                parent[ra] = rb
    return torch.from_numpy(np.array([find(v) for v in range(len(parent))], dtype=labels.numpy().dtype))


def shard_inputs(g, world, rank):
    m = len(g["root"])
    lo, hi = m * rank // world, m * (rank + 1) // world          # split by file chunk
    rec = {k: torch.from_numpy(np.ascontiguousarray(g[k][lo:hi]).astype(
        {"root": np.int64, "ctg": np.int64, "num_pairs": np.int64}.get(k, g[k].dtype)))
        for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
    rec["k"] = torch.arange(lo, hi, dtype=torch.int64)
    return rec


def check_partition(comm, g):
    n = len(g["seq_len"])
    skip = torch.from_numpy((g["astat"] <= 20.0) | (g["copy_num"] < 0.3))
    rec = shard_inputs(g, comm.world, comm.rank)
    labels, rounds = dist_mod.component_labels(comm, n, rec["root"], rec["ctg"], skip, numpy_label_fn, "cpu")
    # reference: union-find over ALL records
    allroot = torch.from_numpy(g["root"].astype(np.int64)); allctg = torch.from_numpy(g["ctg"].astype(np.int64))
    want = numpy_label_fn(torch.arange(n, dtype=torch.int32), allroot, allctg, skip)
    assert labels.dtype == torch.int32 and torch.equal(labels, want)
    owner, load = dist_mod.plan_owners(comm, n, labels, skip, rec["root"], rec["ctg"])
    assert int((owner[~skip] < 0).sum()) == 0 and int((owner[skip] >= 0).sum()) == 0
    assert torch.equal(owner[~skip], owner[labels.to(torch.int64)][~skip])        # one owner per component
    mine = dist_mod.route_records(comm, owner, skip, rec)
    k = mine["k"]
    assert torch.all(k[1:] > k[:-1])                                # file order kept
    a, b = mine["root"], mine["ctg"]
    dest = torch.where(~skip[a], owner[a], torch.where(~skip[b], owner[b], torch.minimum(a, b) % comm.world))
    assert torch.all(dest == comm.rank)
    # every field travels with its record through the packed rows
    for name in ("root", "ctg", "dist", "std_dev", "flags"):
        assert torch.equal(mine[name].to(torch.float64), torch.from_numpy(g[name].astype(np.float64))[k]), name
    assert torch.equal(mine["num_pairs"], torch.from_numpy(g["num_pairs"].astype(np.int64))[k])
    total = torch.tensor([k.numel()], dtype=torch.int64)
    comm.all_reduce(total, "sum")
    assert int(total) == len(g["root"])
    return rounds, load


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = make_inputs(3000, 31, p_repeat=0.05)
        rounds, load = check_partition(dist_mod.TorchComm(), g)
        q.put((rank, rounds, load.tolist()))
    finally:
        dist.destroy_process_group()


def test_partition_over_gloo_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert res[0][2] == res[1][2]                 # same plan on both ranks
    load = res[0][2]
    assert max(load) <= 1.2 * (sum(load) / 2) + 50   # balanced


@pytest.mark.parametrize("world", [2, 3])
def test_partition_in_process(world):
    g = make_inputs(2000, 33, p_repeat=0.05, p_chimeric=0.05)
    shared = dist_mod.ThreadComm.Shared(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            out[r] = check_partition(dist_mod.ThreadComm(shared, r), g)
        except BaseException as ex:   # noqa: B902
            errs.append(ex)
            shared.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_packed_rows_round_trip():
    """32-byte rows: ids up to 2^31-1, negative distances, any float bits, both flags"""
    rng = np.random.default_rng(3)
    k = 5000
    rec = dict(root=torch.from_numpy(rng.integers(0, 2**31 - 1, k)), ctg=torch.from_numpy(rng.integers(0, 2**31 - 1, k)),
               dist=torch.from_numpy(rng.integers(-2**62, 2**62, k)),
               std_dev=torch.from_numpy(rng.standard_normal(k).astype(np.float32) * 1e3),
               num_pairs=torch.from_numpy(rng.integers(0, 2**62, k)),
               flags=torch.from_numpy(rng.integers(0, 4, k).astype(np.uint8)),
               k=torch.from_numpy(rng.integers(0, 2**32 - 1, k)))
    rec["root"][0] = rec["ctg"][1] = 2**31 - 1
    rec["std_dev"][2] = float("inf"); rec["std_dev"][3] = -0.0
    back = dist_mod.unpack_records(dist_mod.pack_records(rec))
    for name, t in rec.items():
        a, b = back[name], t
        if name == "std_dev":
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        else:
            assert torch.equal(a.to(torch.int64), b.to(torch.int64)), name
    empty = dist_mod.unpack_records(dist_mod.pack_records({n_: t[:0] for n_, t in rec.items()}))
    assert all(t.numel() == 0 for t in empty.values())
