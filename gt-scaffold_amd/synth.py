"""Synthetic scaffold-graph inputs (contigs + DistEst records + A-statistics).

The generator models what the reference consumes (ref: testdata/libPE.de,
testdata/libPE.astat, src/gt_scaffolder_parser.c:295-394): contigs laid out on
"true" scaffolds in random orientation, distance estimates between contigs
that lie within the library's reach (each pair listed on the line of BOTH
contigs, as DistanceEst does), repeat contigs with many links and a low
A-statistic, heterozygous "bubble" contigs with copy number ~0.5, and a small
rate of chimeric links (which create inconsistent overlaps and cycles).

It runs on any torch device: CPU for the parity tests (the arrays feed both
the oracle and the HIP engine), cuda for bench.py (100 M records in HBM).
Records are returned in "file order": grouped by root contig in the order of
the contig's line, sense records before antisense ones.
"""
import math

import torch


def _normal(gen, n, device, portable):
    """Standard normal draws.  portable: the sum of twelve uniforms minus six
    (Irwin-Hall), float32 additions only -- IEEE-exact operations, so every host
    draws the same values; the library's normal_() goes through log / cos
    implementations (MKL, libm) whose last bit depends on the CPU."""
    if not portable:
        return torch.empty(n, device=device, dtype=torch.float32).normal_(0.0, 1.0, generator=gen)
    acc = torch.rand(n, device=device, generator=gen)
    for _ in range(11):
        acc = acc + torch.rand(n, device=device, generator=gen)
    return acc - 6.0


def _lognormal_int(gen, n, median, sigma, lo, hi, device, portable=False):
    x = _normal(gen, n, device, portable)
    if portable:   # exp in double, rounded to an integer: one ulp of the library cannot show
        v = (math.log(median) + sigma * x.double()).exp().round().clamp_(lo, hi)
    else:
        v = (math.log(median) + sigma * x).exp().round().clamp_(lo, hi)
    return v.to(torch.int64)


def make_graph(n_contigs, seed=0, device="cpu", links_per_side=5, reach=6000,
               scaffold_median=20, scaffold_sigma=1.0, p_repeat=0.02, repeat_degree=24,
               p_bubble=0.02, p_chimeric=0.01, p_missing_astat=0.01, p_relist=0.01,
               p_link=0.97, contig_median=900, dist_range_small=False, scaffold_max=20000,
               p_relist_flip=0.0, min_dist=-99, p_inversion=1.0, unique_pairs=False,
               p_repeat_unmarked=0.0, portable=False, permute_ids=True, permute_lines=True):
    """Returns a dict of tensors:
      seq_len[u64 as i64], astat[f32], copy_num[f32]          (per contig)
      root[i32], ctg[i32], dist[i64], std_dev[f32], num_pairs[i64], flags[u8]
    flags bit0 = sense, bit1 = same (ref gt_scaffolder_graph.h:63-69).

    portable=True draws the same graph on every host (CPU device): no
    transcendental function reaches an output bit and the file-order sort is
    stable.  It is a different random stream than portable=False; the
    full-size oracle fixture (tools/make_full_size_digest.py) uses it."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n = int(n_contigs)

    def rand(k):
        return torch.rand(k, device=dev, generator=gen)

    def randint(lo, hi, k):
        return torch.randint(lo, hi, (k,), device=dev, generator=gen)

    # --- layout: positions 0..n-1 on concatenated true scaffolds -------
    n_sc = max(4, int(n / scaffold_median * 2.5) + 8)
    sc_len = _lognormal_int(gen, n_sc, scaffold_median, scaffold_sigma, 1, scaffold_max, dev, portable)
    sc_end = torch.cumsum(sc_len, 0)
    while int(sc_end[-1]) < n:  # pathological draw: extend
        more = _lognormal_int(gen, n_sc, scaffold_median, scaffold_sigma, 1, scaffold_max, dev, portable)
        sc_end = torch.cat([sc_end, sc_end[-1] + torch.cumsum(more, 0)])
    pos = torch.arange(n, device=dev)
    sc_id = torch.searchsorted(sc_end, pos, right=True)
    vid = torch.randperm(n, device=dev, generator=gen)  # position -> vertex id
    if not permute_ids:   # measurement aid: contig ids in genome order (what a locality-preserving renumbering would give)
        vid = torch.arange(n, device=dev)

    clen = _lognormal_int(gen, n, contig_median, 0.8, 201, 60000, dev, portable)
    orient = randint(0, 2, n).to(torch.bool)            # True = reverse strand
    gap = randint(-50, 300, n)
    # bubbles: position b and b+1 are two alleles of one locus
    bub = (rand(n) < p_bubble) & (pos + 1 < n)
    bub[1:] &= ~bub[:-1].clone()
    nxt_same = torch.zeros(n, dtype=torch.bool, device=dev)
    nxt_same[:-1] = sc_id[1:] == sc_id[:-1]
    bub &= nxt_same
    gap = torch.where(bub, -clen + randint(0, 20, n), gap)
    step = clen + gap
    cs = torch.cumsum(step, 0) - step                    # global start coordinate
    # (coordinates are only compared inside one scaffold)

    copy_num = (1.0 + 0.08 * _normal(gen, n, dev, portable)).clamp_(0.6, 1.4)
    allele = bub.clone()
    allele[1:] |= bub[:-1]
    copy_num = torch.where(allele, 0.5 + 0.05 * (rand(n) - 0.5), copy_num)
    astat = 25.0 + 4000.0 * rand(n) * (clen.float() / contig_median)
    is_rep = rand(n) < p_repeat
    copy_num = torch.where(is_rep, 2.0 + 6.0 * rand(n), copy_num)
    astat = torch.where(is_rep, -50.0 + 60.0 * rand(n), astat)
    if p_repeat_unmarked > 0:
        # collapsed repeats that look unique (A-statistic and copy number of a
        # single-copy contig): mark_repeats leaves them alone, so their many
        # links reach the filter's hub path and the component programs.  Drawn
        # only when asked for: the random stream of the other workloads stays
        # as it was.
        quiet = is_rep & (rand(n) < p_repeat_unmarked)
        copy_num = torch.where(quiet, 0.9 + 0.2 * rand(n), copy_num)
        astat = torch.where(quiet, 30.0 + 100.0 * rand(n), astat)
    missing = rand(n) < p_missing_astat                  # not in the .astat file
    copy_num = torch.where(missing, torch.zeros_like(copy_num), copy_num)
    astat = torch.where(missing, torch.zeros_like(astat), astat)
    low_cn = rand(n) < 0.003                              # copy number below cut-off
    copy_num = torch.where(low_cn & ~missing, 0.1 + 0.15 * rand(n), copy_num)

    # --- pair list (a = earlier position, b = later position) -----------
    pa, pb, pd = [], [], []
    for k in range(1, links_per_side + 1):
        a = pos[: n - k]
        b = a + k
        d = cs[b] - (cs[a] + clen[a])
        # DistanceEst does not report estimates below -99 (ref src/test.c:49 MIN_DIST)
        ok = (sc_id[a] == sc_id[b]) & (d <= reach) & (d >= min_dist) & (rand(n - k) < p_link)
        ok &= ~(is_rep[a] | is_rep[b])
        if k == 1:
            ok &= ~bub[a]
        pa.append(a[ok]); pb.append(b[ok]); pd.append(d[ok])
    pa = torch.cat(pa); pb = torch.cat(pb); pd = torch.cat(pd)
    n_true = pa.numel()
    # chimeric links: random partner nearby or anywhere, random geometry
    n_chi = int(p_chimeric * n)
    ca = randint(0, n, n_chi)
    near = rand(n_chi) < 0.7
    cb = torch.where(near, (ca + randint(-30, 31, n_chi)).clamp_(0, n - 1), randint(0, n, n_chi))
    keep = ca != cb
    ca, cb = ca[keep], cb[keep]
    if unique_pairs:
        # one estimate per contig pair, as in a real DistEst file: drop a false
        # link that repeats a pair (it would give the pair's two directed edges
        # contradicting geometry)
        tkey = torch.minimum(pa, pb) * n + torch.maximum(pa, pb)
        ckey = torch.minimum(ca, cb) * n + torch.maximum(ca, cb)
        keep = ~torch.isin(ckey, tkey)
        ca, cb, ckey = ca[keep], cb[keep], ckey[keep]
        _, first = torch.unique(ckey, return_inverse=True)
        seen = torch.zeros(int(first.max().item()) + 1 if first.numel() else 1, dtype=torch.bool,
                           device=dev)
        order_c = torch.arange(ckey.numel(), device=dev)
        firstpos = torch.full_like(seen, 0, dtype=torch.int64).scatter_reduce(
            0, first, order_c, reduce="amin", include_self=False) if first.numel() else None
        if first.numel():
            keep = firstpos[first] == order_c
            ca, cb = ca[keep], cb[keep]
    cd = randint(-90, reach, ca.numel())
    # repeat links
    rep_pos = pos[is_rep]
    if portable:
        expo = -torch.log1p(-rand(rep_pos.numel()).double())
    else:
        expo = torch.empty(rep_pos.numel(), device=dev).exponential_(1.0, generator=gen)
    rdeg = (expo * repeat_degree).long().clamp_(1, max(1, n - 1))
    ra = torch.repeat_interleave(rep_pos, rdeg)
    rb = randint(0, n, ra.numel())
    keep = ra != rb
    ra, rb = ra[keep], rb[keep]
    rd = randint(-90, reach, ra.numel())

    A = torch.cat([pa, ca, ra]); B = torch.cat([pb, cb, rb]); D = torch.cat([pd, cd, rd])
    m = A.numel()
    # geometry flags.  true links follow the layout; others are random.
    sense_a = ~orient[A]                      # B lies downstream of A
    same = orient[A] == orient[B]
    # false links: repeat links always get random geometry; a chimeric link gets
    # random geometry (an inversion: it joins the two strands of the layout)
    # with probability p_inversion, else the geometry of the layout with a
    # wrong position / distance (a plain misjoin).  Inversions put a vertex on
    # walks in both directions; the reference's label-correcting walk search
    # (algorithms.c:681-728) then degenerates (millions of queue pops for a
    # few hundred contigs).
    rnd = torch.ones(m, dtype=torch.bool, device=dev)
    rnd[:n_true] = False
    n_c = ca.numel()
    rnd[n_true:n_true + n_c] = rand(n_c) < p_inversion
    sense_a = torch.where(rnd, rand(m) < 0.5, sense_a)
    same = torch.where(rnd, rand(m) < 0.5, same)
    sense_b = torch.where(same, ~sense_a, sense_a)  # ref parser.c:369-372 twin_dir

    npairs = randint(5, 600, m)
    sigma_lib = 60.0
    if portable:   # the square root in double; only its value rounded to 0.1 is kept
        sd = (sigma_lib / npairs.double().sqrt() * (0.8 + 0.4 * rand(m)).double())
        sd = ((sd * 10).round() / 10).float()
    else:
        sd = (sigma_lib / npairs.float().sqrt() * (0.8 + 0.4 * rand(m)))
        sd = (sd * 10).round() / 10                       # .de files carry %.1f
    noise = (_normal(gen, m, dev, portable) * sd).round().long()
    dist = (D + noise).clamp_(min=min_dist)
    if dist_range_small:                                  # provoke ties in walks
        dist = dist.clamp_(-99, 99)

    # records: one on each contig's line; a few pairs are listed again with
    # another estimate (exercises the alter-edge rule, ref parser.c:362-366)
    VA, VB = vid[A], vid[B]
    root = torch.cat([VA, VB]); ctg = torch.cat([VB, VA])
    rsense = torch.cat([sense_a, sense_b]); rsame = torch.cat([same, same])
    rdist = torch.cat([dist, dist]); rsd = torch.cat([sd, sd]); rnp = torch.cat([npairs, npairs])
    n_re = int(p_relist * m)
    if n_re > 0:
        idx = randint(0, 2 * m, n_re)
        root = torch.cat([root, root[idx]]); ctg = torch.cat([ctg, ctg[idx]])
        # a re-listed pair normally keeps its direction; p_relist_flip makes the two
        # directed edges of a pair geometrically inconsistent (with a negative
        # distance this is a negative 2-cycle on which the reference's walk
        # search does not terminate, so it is off by default)
        rsense = torch.cat([rsense, torch.where(rand(n_re) < 1.0 - p_relist_flip, rsense[idx], ~rsense[idx])])
        rsame = torch.cat([rsame, rsame[idx]])
        rdist = torch.cat([rdist, rdist[idx] + randint(-40, 41, n_re)])
        rsd = torch.cat([rsd, ((rsd[idx] + 0.6 * (rand(n_re) - 0.4)) * 10).round().clamp_(1, 500) / 10])
        rnp = torch.cat([rnp, randint(5, 600, n_re)])
    # file order: line order of the root (a random contig order), sense first
    line_of = torch.randperm(n, device=dev, generator=gen)
    if not permute_lines:   # measurement aid: the roots' lines in contig-id order
        line_of = torch.arange(n, device=dev)
    key = line_of[root].to(torch.int64) * 2 + (~rsense).to(torch.int64)
    key = key * (1 << 22) + randint(0, 1 << 22, key.numel())
    order = torch.argsort(key, stable=True) if portable else torch.argsort(key)
    flags = (rsense.to(torch.uint8) | (rsame.to(torch.uint8) << 1))

    out = dict(
        seq_len=torch.empty(n, dtype=torch.int64, device=dev),
        astat=torch.empty(n, dtype=torch.float32, device=dev),
        copy_num=torch.empty(n, dtype=torch.float32, device=dev),
        root=root[order].to(torch.int32).contiguous(),
        ctg=ctg[order].to(torch.int32).contiguous(),
        dist=rdist[order].contiguous(),
        std_dev=rsd[order].to(torch.float32).contiguous(),
        num_pairs=rnp[order].contiguous(),
        flags=flags[order].contiguous(),
    )
    out["seq_len"][vid] = clen
    out["astat"][vid] = astat.to(torch.float32)
    out["copy_num"][vid] = copy_num.to(torch.float32)
    return out


def to_numpy(g):
    import numpy as np
    o = {k: v.cpu().numpy() for k, v in g.items()}
    o["seq_len"] = o["seq_len"].astype(np.uint64)
    o["root"] = o["root"].astype(np.uint32)
    o["ctg"] = o["ctg"].astype(np.uint32)
    o["num_pairs"] = o["num_pairs"].astype(np.uint64)
    return o


def header_of(i, width=9):
    """Synthetic contig names sort lexicographically in numeric order, so the
    vertex id the reference assigns (qsort by header, ref parser.c:172) is i."""
    return "contig-%0*d" % (width, i)


def write_files(g, prefix, fasta_order_seed=1):
    """Write <prefix>.fa / .de / .astat in the formats the reference parses.
    g: numpy dict from to_numpy().  Contigs absent from the .astat file are
    those with astat == 0 and copy_num == 0."""
    import numpy as np
    n = len(g["seq_len"])
    rng = np.random.default_rng(fasta_order_seed)
    with open(prefix + ".fa", "w") as f:
        for i in rng.permutation(n):
            L = int(g["seq_len"][i])
            f.write(">%s %d 0\n" % (header_of(i), L))
            s = "ACGT" * (L // 4 + 1)
            s = s[:L]
            for o in range(0, L, 60):
                f.write(s[o:o + 60] + "\n")
    with open(prefix + ".astat", "w") as f:
        for i in range(n):
            if g["astat"][i] == 0 and g["copy_num"][i] == 0:
                continue
            f.write("%s\t%d\t%d\t%d\t%s\t%s\n" % (header_of(i), g["seq_len"][i], 0, 0,
                                                   repr(float(g["copy_num"][i])),
                                                   repr(float(g["astat"][i]))))
    with open(prefix + ".de", "w") as f:
        # one line per root contig: sense records, ';', antisense records.  The
        # reference reads lines into a 1024-byte buffer (parser.c:30), so long
        # lists continue on a further line of the same root.
        root = g["root"]; m = len(root)
        k = 0
        seen = set()
        while k < m:
            r = int(root[k])
            parts = [header_of(r)]
            length = len(parts[0])
            sense = True
            while k < m and int(root[k]) == r:
                s = bool(g["flags"][k] & 1)
                rec = "%s%s,%d,%d,%s" % (header_of(int(g["ctg"][k])),
                                         "+" if g["flags"][k] & 2 else "-",
                                         g["dist"][k], g["num_pairs"][k],
                                         repr(float(g["std_dev"][k])))
                if (not sense) and s:         # back to sense: needs a new line
                    break
                if length + len(rec) + 4 > 900:
                    break
                if sense and not s:
                    parts.append(";")
                    sense = False
                    length += 2
                parts.append(rec)
                length += len(rec) + 1
                k += 1
            if sense:
                parts.append(";")
            seen.add(r)
            f.write(" ".join(parts) + "\n")
        for i in range(n):
            if i not in seen:
                f.write("%s ;\n" % header_of(i))
