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
  3. route  : ONE all_to_all of the records, packed into 32 bytes each, to the
              owner of their component (a record touching a repeat contig
              follows its other contig).

After routing a rank renumbers its contigs -- the ones it owns plus the repeat
contigs, which every shard shares -- with consecutive local numbers in the order
of their ids, so every vertex-indexed kernel of the engine runs over
n / world + repeats vertices.  The engine gets the whole-graph id of every local
vertex as its time stamp (gtsg_set_vertex_times): the only effect that crosses
shards is the time of the latest inconsistency hit on the edges of the repeat
contigs (ref algorithms.c:249-258), one all_reduce MAX over 2 x int32 per
repeat contig between the two halves of the filter.

Collectives run through a `comm` object so that the same code is driven by
torch.distributed (RCCL on GPUs, gloo in the CPU tests) or by the in-process
communicator the single-GPU test uses.
"""
import threading

import torch

ROW_WORDS = 4   # int64 words of a packed record


class TorchComm:
    """torch.distributed communicator (backend nccl = RCCL on ROCm, or gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._on_host = dist.get_backend(group) != "nccl"   # gloo moves CPU tensors

    def all_reduce(self, t, op):
        d = self.dist
        rop = {"min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX, "sum": d.ReduceOp.SUM}[op]
        if self._on_host and t.is_cuda:
            h = t.cpu()
            d.all_reduce(h, op=rop, group=self.group)
            t.copy_(h)
        else:
            d.all_reduce(t, op=rop, group=self.group)
        return t

    def exchange_rows(self, rows, counts):
        """rows: [k, W] tensor grouped by destination rank, counts[r] rows for
        rank r.  Returns the rows received, grouped by source rank."""
        d = self.dist
        dev = rows.device
        if self._on_host:
            rows = rows.cpu()
        cnt = torch.tensor(counts, dtype=torch.int64, device=rows.device)
        rcnt = torch.empty_like(cnt)
        d.all_to_all_single(rcnt, cnt, group=self.group)
        rc = rcnt.tolist()
        out = torch.empty((sum(rc), rows.shape[1]), dtype=rows.dtype, device=rows.device)
        d.all_to_all_single(out, rows.contiguous(), rc, list(counts), group=self.group)
        return out.to(dev)


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
        if self.world == 1:
            return t
        s.slots[self.rank] = t.clone()
        s.barrier.wait()
        st = torch.stack([x.to(t.device) for x in s.slots])
        r = {"min": lambda: st.min(0).values, "max": lambda: st.max(0).values,
             "sum": lambda: st.sum(0)}[op]()
        s.barrier.wait()
        t.copy_(r)
        return t

    def exchange_rows(self, rows, counts):
        s = self.s
        if self.world == 1:
            return rows
        s.slots[self.rank] = list(torch.split(rows, list(counts)))
        s.barrier.wait()
        out = torch.cat([s.slots[r][self.rank].clone() for r in range(self.world)])
        s.barrier.wait()
        return out


# ---- packed records -------------------------------------------------------
def pack_records(rec):
    """rec: root, ctg (ids below 2^31), dist (i64), std_dev (f32), num_pairs
    (i64), flags (u8: bit0 sense, bit1 same), k (global record index = file
    order, below 2^32) -> int64 [n, 4]:
      word 0: root | sense << 31 | ctg << 32 | same << 63
      word 1: dist        word 2: num_pairs
      word 3: k | bits(std_dev) << 32"""
    i64 = torch.int64
    fl = rec["flags"].to(i64)
    w0 = rec["root"].to(i64) | ((fl & 1) << 31) | (rec["ctg"].to(i64) << 32) | (((fl >> 1) & 1) << 63)
    sd = rec["std_dev"].contiguous().view(torch.int32).to(i64) & 0xFFFFFFFF
    w3 = rec["k"].to(i64) | (sd << 32)
    return torch.stack([w0, rec["dist"].to(i64), rec["num_pairs"].to(i64), w3], dim=1)


def unpack_records(rows):
    w0, w3 = rows[:, 0], rows[:, 3]
    sd = ((w3 >> 32) & 0xFFFFFFFF).to(torch.int32).view(torch.float32) if rows.numel() else \
        torch.empty(0, dtype=torch.float32, device=rows.device)
    return dict(root=w0 & 0x7FFFFFFF, ctg=(w0 >> 32) & 0x7FFFFFFF,
                flags=(((w0 >> 31) & 1) | (((w0 >> 63) & 1) << 1)).to(torch.uint8),
                dist=rows[:, 1].contiguous(), num_pairs=rows[:, 2].contiguous(),
                k=w3 & 0xFFFFFFFF, std_dev=sd.contiguous())


# ---- the three steps --------------------------------------------------------
def component_labels(comm, n, root, ctg, skip, label_fn, device, force_collectives=False):
    """Step 1.  root / ctg: this shard's records, skip: bool[n] (repeat
    contigs).  Returns labels[n] (int32): smallest contig of the component,
    identical on every rank.
    A round: join the contigs of the local records (label_fn runs its union-find
    to the end), then the element-wise MIN over the shards.  The fixpoint is
    reached when a round changes nothing on any shard; the "changed" flag rides
    in one extra element of the reduced tensor (-1 = changed: MIN), so a round is
    one collective and one look at the host.  One shard: the first round's
    labels are final."""
    buf = torch.empty(n + 1, dtype=torch.int32, device=device)
    labels = buf[:n]
    torch.arange(n, dtype=torch.int32, device=device, out=labels)
    rounds = 0
    while True:
        many = comm.world > 1 or force_collectives
        prev = labels.clone() if many else None
        out = label_fn(labels, root, ctg, skip)
        if out.data_ptr() != labels.data_ptr():
            labels.copy_(out)
        rounds += 1
        if not many:
            return labels, rounds
        buf[n] = -(labels != prev).any().to(torch.int32)
        comm.all_reduce(buf, "min")
        if int(buf[n].item()) == 0:
            return labels, rounds


def plan_owners(comm, n, labels, skip, root, ctg):
    """Step 2.  Returns owner[n] (int64 rank of every contig, -1 for the shared
    repeat contigs) and the per-rank record weight of the plan."""
    dev = labels.device
    labels = labels.to(torch.int64)
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


def plan_owners_engine(comm, eng, labels, skip8, root32, ctg32):
    """Step 2 with the engine's kernels (gtsg_plan_weights / gtsg_plan_deal):
    nothing of the plan is computed in torch ops and nothing is read back on the
    way -- count, ONE all_reduce of the weights (int32 per contig), sort + deal.
    Returns (owner int8 [n], load int64 [world])."""
    w = eng.plan_weights(root32, ctg32, skip8, labels)
    comm.all_reduce(w, "sum")
    return eng.plan_deal(skip8, labels, w, comm.world)


def route_records_engine(comm, eng, owner, rec, loc_of):
    """Step 3 with the engine's HIP kernels (gtsg_route_pack / _unpack): the
    destination of every record, one stable 8-bit sort pass by destination and
    the 32-byte rows in one sweep; after the all_to_all the rows are unpacked
    straight into the shard's local contig numbers.  rec["k"] must be a run of
    consecutive indices (a chunk of the record file)."""
    k = rec["k"]
    first = int(k[0]) if k.numel() else 0
    r = dict(root=rec["root"].to(torch.int32).contiguous(), ctg=rec["ctg"].to(torch.int32).contiguous(),
             dist=rec["dist"].to(torch.int64).contiguous(), std_dev=rec["std_dev"].to(torch.float32).contiguous(),
             num_pairs=rec["num_pairs"].to(torch.int64).contiguous(), flags=rec["flags"].to(torch.uint8).contiguous())
    rows, counts = eng.route_pack(r, first, owner if owner.dtype == torch.int8 else owner.to(torch.int8), comm.world)
    rows = comm.exchange_rows(rows, counts)
    out = eng.route_unpack(rows.contiguous(), loc_of)
    if out.pop("out_of_order"):            # chunks dealt in file order arrive sorted: the kernel looks
        o = torch.argsort(out["k"], stable=True)
        out = {name: t[o] for name, t in out.items()}
    return out


def route_records(comm, owner, skip, rec):
    """Step 3 in torch (CPU tests, or records that are not a chunk of the
    file).  rec: dict of equally long 1-D tensors (root, ctg, dist, std_dev,
    num_pairs, flags, k = global record index).  One packed all_to_all; returns
    this rank's records sorted by k (= file order)."""
    root, ctg = rec["root"].to(torch.int64), rec["ctg"].to(torch.int64)
    # both contigs repeats: any rank, but the same one for every record of the pair
    dest = torch.where(~skip[root], owner[root],
                       torch.where(~skip[ctg], owner[ctg], torch.minimum(root, ctg) % comm.world))
    counts = torch.bincount(dest, minlength=comm.world).tolist()
    rows = pack_records(rec)
    if comm.world > 1:
        # stable, so the file order survives inside a destination; one-byte keys
        # (a rank number) sort in a single radix pass
        order = torch.sort(dest.to(torch.uint8 if comm.world <= 256 else torch.int32), stable=True)[1]
        rows = rows[order]
    rows = comm.exchange_rows(rows, counts)
    out = unpack_records(rows)
    k = out["k"]
    if k.numel() > 1 and not bool((k[1:] >= k[:-1]).all()):   # chunks dealt in file order arrive sorted
        o = torch.argsort(k, stable=True)
        out = {name: t[o] for name, t in out.items()}
    return out


def engine_label_fn(eng):
    """label_fn backed by the engine's HIP kernels (gtsg_label_components); the
    records and the skip mask are converted once, not per round."""
    cache = {}

    def fn(labels, root, ctg, skip):
        key = (root.data_ptr(), ctg.data_ptr(), skip.data_ptr())
        if cache.get("key") != key:
            cache.update(key=key, root=root.to(torch.int32).contiguous(), ctg=ctg.to(torch.int32).contiguous(),
                         skip=skip.to(torch.uint8).contiguous())
        lab = labels.contiguous()
        eng.label_components(lab.numel(), cache["root"], cache["ctg"], cache["skip"], lab)
        return lab
    return fn


def scaffold_sharded(comm, eng, contigs, rec, cuts, label_fn=None, timers=None, force_collectives=False):
    """The whole hot path for one graph whose records are split over the
    ranks.  contigs: seq_len / astat / copy_num of ALL contigs (tensors on the
    engine's device); rec: this rank's slice of the records with global index
    k.  Returns (owner[n], labelling rounds, plan load, local[n_local]): the
    engine holds the shard with LOCAL vertex numbers, local[v] is the contig id
    of local vertex v (owned contigs and all repeat contigs, ascending); vertex
    states are valid for owned and repeat contigs, every edge lives on exactly
    one rank.  timers: optional dict that receives the wall time (s) of the
    stages, each closed by a device synchronisation (measurement only).
    force_collectives: run the collectives a single rank would skip (the MIN of
    the labels, the MAX of the latest hits) -- the one-rank RCCL test."""
    import time
    t_last = [time.perf_counter()]

    def lap(name):
        if timers is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            now = time.perf_counter()
            timers[name] = timers.get(name, 0.0) + now - t_last[0]
            t_last[0] = now
    dev = contigs["seq_len"].device
    n = contigs["seq_len"].numel()
    skip = (contigs["astat"] <= cuts["astat_cutoff"]) | (contigs["copy_num"] < cuts["copy_num_cutoff"])
    on_engine = dev.type == "cuda" and label_fn is None and hasattr(eng, "plan_weights")
    if on_engine:
        # ids stay 32 bit on the device; the engine's kernels take them as they are
        root = rec["root"] if rec["root"].dtype == torch.int32 else rec["root"].to(torch.int32)
        ctg = rec["ctg"] if rec["ctg"].dtype == torch.int32 else rec["ctg"].to(torch.int32)
        skip8 = skip.to(torch.uint8)
        labels, rounds = component_labels(comm, n, root.contiguous(), ctg.contiguous(), skip8,
                                          lambda lab, r, c, s: (eng.label_components(lab.numel(), r, c, s, lab), lab)[1],
                                          dev, force_collectives)
        lap("label")
        owner, load = plan_owners_engine(comm, eng, labels, skip8, root, ctg)
    else:
        root, ctg = rec["root"].to(torch.int64), rec["ctg"].to(torch.int64)
        labels, rounds = component_labels(comm, n, root, ctg, skip, label_fn or engine_label_fn(eng), dev,
                                          force_collectives)
        lap("label")
        owner, load = plan_owners(comm, n, labels, skip, root, ctg)
    lap("plan")
    # local numbering: owned + repeat contigs, in id order
    member = (owner == comm.rank) | skip
    local = torch.nonzero(member).flatten()
    loc_of = torch.cumsum(member.to(torch.int32), 0, dtype=torch.int32) - 1
    k = rec["k"]
    chunk = k.numel() == 0 or int(k[-1]) - int(k[0]) + 1 == k.numel()
    if dev.type == "cuda" and chunk and hasattr(eng, "route_pack"):
        mine = route_records_engine(comm, eng, owner, rec, loc_of)
    else:
        mine = route_records(comm, owner, skip, rec)
        mine["root"], mine["ctg"] = loc_of[mine["root"]], loc_of[mine["ctg"]]
    lap("route")
    eng.set_contigs(contigs["seq_len"][local].contiguous(), contigs["astat"][local].contiguous(),
                    contigs["copy_num"][local].contiguous())
    eng.set_vertex_times(local.to(torch.int32).contiguous())
    eng.build_from_records(mine["root"].to(torch.int32).contiguous(),
                           mine["ctg"].to(torch.int32).contiguous(),
                           mine["dist"].contiguous(), mine["std_dev"].contiguous(),
                           mine["num_pairs"].contiguous(), mine["flags"].contiguous())
    eng.mark_repeats(True, cuts["copy_num_cutoff"], cuts["astat_cutoff"])
    lap("renumber_build_mark")
    # latest-hit times of the repeat contigs' edges: whole-graph ids, MAX over
    # the shards (every rank holds all repeat contigs, so a rank without any
    # vertex means there is no repeat and nothing to combine)
    rep_loc = loc_of[torch.nonzero(skip).flatten()].to(torch.int64)
    if local.numel():
        eng.filter_begin(cuts["pcutoff"], cuts["cncutoff"], cuts["ocutoff"])
        if comm.world > 1 or force_collectives:
            lasthit = torch.empty(2 * local.numel(), dtype=torch.int32, device=dev)
            eng.filter_get_lasthit(lasthit)
    if rep_loc.numel() and (comm.world > 1 or force_collectives):     # (one shard has nobody to agree with)
        tab = lasthit.view(-1, 2)[rep_loc].contiguous()
        comm.all_reduce(tab, "max")
        lasthit.view(-1, 2)[rep_loc] = tab
        eng.filter_set_lasthit(lasthit)
    if local.numel():
        eng.filter_end()
        lap("filter")
        eng.makescaffold()
        lap("makescaffold")
    return owner, rounds, load, local
