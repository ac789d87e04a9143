"""Multi-GPU scaffolding of ONE graph: the component-partition step and the
sharded pipeline ('Multi-GPU').

The hot path shards by connected component: once repeat contigs are marked,
everything the reference does (filter, cycle removal, walks) stays inside a
connected component of the unmarked contigs.  Records arrive split by file
chunk, so the shards first agree on the components and then move every record
to the GPU that owns its component:

  1. label  : each shard joins the contigs of its records (engine kernel
              gtsg_label_components, or any `label_fn`), the shards take the
              element-wise MIN of the parent arrays (all_reduce over RCCL /
              xGMI, 4 B per contig) and repeat until nothing changes;
  2. plan   : records per component, summed over the shards (all_reduce SUM),
              give every component to a rank, largest first in serpentine
              order -- computed identically on every rank;
  3. route  : all_to_all of the records to the owner of their component (a
              record touching a repeat contig follows its other contig).

Repeat contigs are shared by the shards.  The only effect that crosses shards
is the time of the latest inconsistency hit on their edges
(ref algorithms.c:249-258): one all_reduce MAX between the two halves of the
filter.  Collectives run through a `comm` object so that the same code is
driven by torch.distributed (RCCL on GPUs, gloo in the CPU tests) or by the
in-process communicator the single-GPU test uses.
"""
import threading

import torch


class TorchComm:
    """torch.distributed communicator (backend nccl = RCCL on ROCm, or gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._a2a = dist.get_backend(group) == "nccl"

    def all_reduce(self, t, op):
        d = self.dist
        d.all_reduce(t, op={"min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX,
                            "sum": d.ReduceOp.SUM}[op], group=self.group)
        return t

    def exchange(self, send):
        """send[r] = 1-D tensor for rank r; returns the tensors received."""
        d = self.dist
        if self._a2a:
            cnt = torch.tensor([x.numel() for x in send], dtype=torch.int64, device=send[0].device)
            rcnt = torch.empty_like(cnt)
            d.all_to_all_single(rcnt, cnt, group=self.group)
            rc = rcnt.tolist()
            out = torch.empty(sum(rc), dtype=send[0].dtype, device=send[0].device)
            d.all_to_all_single(out, torch.cat(send), rc, cnt.tolist(), group=self.group)
            return list(torch.split(out, rc))
        # gloo has no all_to_all: gather everything, keep what is addressed to us
        box = [None] * self.world
        d.all_gather_object(box, [x.cpu() for x in send], group=self.group)
        return [box[r][self.rank].to(send[0].device) for r in range(self.world)]


class ThreadComm:
    """In-process communicator: `world` threads of one process play the ranks
    (single-GPU rehearsal of the sharded pipeline, and unit tests)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def all_reduce(self, t, op):
        s = self.s
        s.slots[self.rank] = t.clone()
        s.barrier.wait()
        st = torch.stack([x.to(t.device) for x in s.slots])
        r = {"min": lambda: st.min(0).values, "max": lambda: st.max(0).values,
             "sum": lambda: st.sum(0)}[op]()
        s.barrier.wait()
        t.copy_(r)
        return t

    def exchange(self, send):
        s = self.s
        s.slots[self.rank] = send
        s.barrier.wait()
        out = [s.slots[r][self.rank].clone() for r in range(self.world)]
        s.barrier.wait()
        return out


def component_labels(comm, n, root, ctg, skip, label_fn, device):
    """Step 1.  root / ctg: this shard's records (int64 tensors), skip: bool[n]
    (repeat contigs).  Returns labels[n] (int64): smallest contig of the
    component, identical on every rank."""
    labels = torch.arange(n, dtype=torch.int64, device=device)
    rounds = 0
    while True:
        prev = labels.clone()
        labels = label_fn(labels, root, ctg, skip)
        comm.all_reduce(labels, "min")
        changed = (labels != prev).any().to(torch.int64).reshape(1)
        comm.all_reduce(changed, "max")
        rounds += 1
        if not int(changed.item()):
            return labels, rounds


def plan_owners(comm, n, labels, skip, root, ctg):
    """Step 2.  Returns owner[n] (int64 rank of every contig, -1 for the shared
    repeat contigs) and the per-rank record weight of the plan."""
    dev = labels.device
    a_ok, b_ok = ~skip[root], ~skip[ctg]
    anchor = torch.where(a_ok, root, ctg)
    keep = a_ok | b_ok
    w = torch.zeros(n, dtype=torch.int64, device=dev)
    w.index_add_(0, labels[anchor[keep]], torch.ones(int(keep.sum()), dtype=torch.int64, device=dev))
    comm.all_reduce(w, "sum")
    roots = torch.nonzero((labels == torch.arange(n, device=dev)) & ~skip).flatten()
    order = torch.argsort(w[roots], descending=True, stable=True)
    pos = torch.arange(roots.numel(), device=dev)
    lap, col = pos // comm.world, pos % comm.world
    rank_of = torch.where(lap % 2 == 0, col, comm.world - 1 - col)
    owner_of_root = torch.full((n,), -1, dtype=torch.int64, device=dev)
    owner_of_root[roots[order]] = rank_of
    owner = torch.where(skip, torch.full_like(labels, -1), owner_of_root[labels])
    load = torch.zeros(comm.world, dtype=torch.int64, device=dev)
    load.index_add_(0, rank_of, w[roots[order]])
    return owner, load


def route_records(comm, owner, skip, rec):
    """Step 3.  rec: dict of equally long 1-D tensors with keys root, ctg, k
    (global record index = file order) and any payload.  Returns this rank's
    records, sorted by k."""
    root, ctg, k = rec["root"], rec["ctg"], rec["k"]
    # both contigs repeats: any rank, but the same one for every record of the pair
    dest = torch.where(~skip[root], owner[root],
                       torch.where(~skip[ctg], owner[ctg], torch.minimum(root, ctg) % comm.world))
    out = {}
    sel = [torch.nonzero(dest == r).flatten() for r in range(comm.world)]
    for name, t in rec.items():
        out[name] = torch.cat(comm.exchange([t[i] for i in sel]))
    o = torch.argsort(out["k"], stable=True)
    return {name: t[o] for name, t in out.items()}


def engine_label_fn(eng):
    """label_fn backed by the engine's HIP kernels (gtsg_label_components)."""
    def fn(labels, root, ctg, skip):
        lab = labels.to(torch.int32).contiguous()
        eng.label_components(lab.numel(), root.to(torch.int32).contiguous(),
                             ctg.to(torch.int32).contiguous(),
                             skip.to(torch.uint8).contiguous(), lab)
        return lab.to(torch.int64)
    return fn


def scaffold_sharded(comm, eng, contigs, rec, cuts, label_fn=None):
    """The whole hot path for one graph whose records are split over the
    ranks.  contigs: seq_len / astat / copy_num (replicated, tensors on the
    engine's device); rec: this rank's slice of the records with global index
    k.  Returns owner[n] and the number of labelling rounds; results stay in
    `eng` (vertex states are valid for owned and repeat contigs, every edge
    lives on exactly one rank)."""
    dev = contigs["seq_len"].device
    n = contigs["seq_len"].numel()
    skip = (contigs["astat"] <= cuts["astat_cutoff"]) | (contigs["copy_num"] < cuts["copy_num_cutoff"])
    root, ctg = rec["root"].to(torch.int64), rec["ctg"].to(torch.int64)
    labels, rounds = component_labels(comm, n, root, ctg, skip, label_fn or engine_label_fn(eng), dev)
    owner, load = plan_owners(comm, n, labels, skip, root, ctg)
    rec64 = dict(rec)
    rec64["root"], rec64["ctg"] = root, ctg
    mine = route_records(comm, owner, skip, rec64)
    eng.set_contigs(contigs["seq_len"], contigs["astat"], contigs["copy_num"])
    eng.build_from_records(mine["root"].to(torch.int32).contiguous(),
                           mine["ctg"].to(torch.int32).contiguous(),
                           mine["dist"].contiguous(), mine["std_dev"].contiguous(),
                           mine["num_pairs"].contiguous(), mine["flags"].contiguous())
    eng.mark_repeats(True, cuts["copy_num_cutoff"], cuts["astat_cutoff"])
    eng.filter_begin(cuts["pcutoff"], cuts["cncutoff"], cuts["ocutoff"])
    lasthit = torch.empty(2 * n, dtype=torch.int32, device=dev)
    eng.filter_get_lasthit(lasthit)
    comm.all_reduce(lasthit, "max")
    eng.filter_set_lasthit(lasthit)
    eng.filter_end()
    eng.makescaffold()
    return owner, rounds, load
