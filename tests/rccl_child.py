"""Child process of test_sharded_pipeline_over_rccl_one_rank: the sharded pipeline through
TorchComm on the nccl (= RCCL) backend with ONE rank, so that every collective and dtype of
gt-scaffold_amd/dist.py -- all_reduce MIN int32[n+1], SUM int32[n], MAX int32[k,2],
all_to_all_single int64[k,4] with split sizes -- runs on RCCL once, compared with the oracle.
The process group is initialised before anything touches the GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29517")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import make_inputs, oracle_from_inputs, pkg  # noqa: E402

g = make_inputs(6000, 41, p_repeat=0.05, p_chimeric=0.05)
n, m = len(g["seq_len"]), len(g["root"])
og = oracle_from_inputs(g)
og.mark_repeats(); og.filter(); og.makescaffold(True)
cuts = dict(copy_num_cutoff=0.3, astat_cutoff=20.0, pcutoff=0.01, cncutoff=1.5, ocutoff=400)
dev = "cuda:0"
eng = pkg.engine.Engine(0)
contigs = dict(seq_len=torch.from_numpy(g["seq_len"].astype(np.int64)).to(dev),
               astat=torch.from_numpy(g["astat"]).to(dev), copy_num=torch.from_numpy(g["copy_num"]).to(dev))
rec = {k: torch.from_numpy(np.ascontiguousarray(g[k]).astype(
    {"root": np.int64, "ctg": np.int64, "num_pairs": np.int64}.get(k, g[k].dtype))).to(dev)
    for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
rec["k"] = torch.arange(0, m, dtype=torch.int64, device=dev)
comm = pkg.dist.TorchComm()
assert comm.world == 1 and not comm._on_host
calls = []
_ar, _ex = comm.all_reduce, comm.exchange_rows
comm.all_reduce = lambda t, op: (calls.append((op, str(t.dtype), tuple(t.shape))), _ar(t, op))[1]
comm.exchange_rows = lambda rows, counts: (calls.append(("all_to_all", str(rows.dtype), tuple(rows.shape))), _ex(rows, counts))[1]
owner, rounds, load, local = pkg.dist.scaffold_sharded(comm, eng, contigs, rec, cuts, force_collectives=True)
torch.cuda.synchronize()
assert local.numel() == n and bool((local == torch.arange(n, device=dev)).all())
assert np.array_equal(eng.vertex_states(), og.vertex_states()), "vertex states"
oe, ee = og.edges(), eng.edges()
want = {(int(a), int(b)): int(s) for a, b, s in zip(oe["start"], oe["end"], og.edge_states())}
got = {(int(a), int(b)): int(s) for a, b, s in zip(ee["start"], ee["end"], eng.edge_states())}
assert got == want, "edge states"
ops = {c[0] for c in calls}
assert ops == {"min", "sum", "max", "all_to_all"}, calls
print("rccl one rank ok:", sorted(ops), "rounds", rounds)
dist.destroy_process_group()
