"""Parity of the HIP engine (through the C ABI) with the CPU oracle, stage by
stage, on seeded synthetic graphs and on the reference's own test data.
Integer / state work: bit-exact."""
import numpy as np
import pytest

from helpers import DEFAULTS, make_inputs, oracle_from_inputs, pkg
from oracle.oracle_py import OracleGraph, lib as oracle_lib

pytestmark = pytest.mark.gpu


def engine_from_inputs(g, **opts):
    eng = pkg.engine.Engine(0)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"],
                           g["num_pairs"].astype(np.int64), g["flags"])
    return eng


def assert_same_graph(eng, og):
    assert eng.nv == og.nv and eng.ne == og.ne, (eng.nv, og.nv, eng.ne, og.ne)
    a, b = eng.edges(), og.edges()
    for k in ("start", "end", "dist", "std_dev", "flags"):
        assert np.array_equal(a[k], b[k]), "edge field %s differs at %s" % (k, np.nonzero(a[k] != b[k])[0][:8])
    assert np.array_equal(a["num_pairs"].astype(np.uint64), b["num_pairs"])


def assert_same_states(eng, og, tag):
    ev, ov = eng.vertex_states(), og.vertex_states()
    assert np.array_equal(ev, ov), "%s: vertex states differ at %s" % (tag, np.nonzero(ev != ov)[0][:8])
    ee, oe = eng.edge_states(), og.edge_states()
    assert np.array_equal(ee, oe), "%s: edge states differ at %s" % (tag, np.nonzero(ee != oe)[0][:8])


def run_pipeline(g, pcutoff=0.01, cncutoff=1.5, ocutoff=400, **opts):
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g, **opts)
    assert_same_graph(eng, og)
    og.mark_repeats(); eng.mark_repeats()
    assert_same_states(eng, og, "mark_repeats")
    og.filter(pcutoff, cncutoff, ocutoff); eng.filter(pcutoff, cncutoff, ocutoff)
    assert_same_states(eng, og, "filter")
    og.removecycles(); eng.removecycles()
    assert_same_states(eng, og, "removecycles")
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold")
    return eng, og


def test_device_rounding_of_ambiguous_order():
    rng = np.random.default_rng(0)
    n = 400000
    d1 = rng.integers(-5000, 5000, n); d2 = d1 + rng.integers(-40, 40, n)
    big = rng.random(n) < 0.1            # int64 -> float rounding
    d1 = np.where(big, d1 * (1 << 40) + rng.integers(0, 1 << 30, n), d1)
    s1 = (rng.random(n) * 30).astype(np.float32); s2 = (rng.random(n) * 30).astype(np.float32)
    s1[rng.random(n) < 0.01] = 0; s2[rng.random(n) < 0.01] = 0
    eng = pkg.engine.Engine(0)
    L = oracle_lib()
    for pc in (0.01, 0.2, 0.0):
        got = eng.selftest_ambiguous(d1, s1, d2, s2, pc)
        want = np.array([L.ora_ambiguousorder(int(a), float(b), int(c), float(d), pc)
                         for a, b, c, d in zip(d1[:60000], s1[:60000], d2[:60000], s2[:60000])], np.uint8)
        assert np.array_equal(got[:60000], want)


@pytest.mark.parametrize("seed", range(6))
def test_small_graphs_stage_by_stage(seed):
    g = make_inputs(400 + 211 * seed, seed, p_chimeric=0.03, p_bubble=0.05, p_repeat=0.03,
                    p_relist_flip=0.1)
    run_pipeline(g)


@pytest.mark.parametrize("seed", range(3))
def test_noisy_graphs(seed):
    g = make_inputs(3000, 100 + seed, p_chimeric=0.3, p_bubble=0.1, p_repeat=0.05, links_per_side=3)
    run_pipeline(g)


@pytest.mark.parametrize("ocutoff,pcutoff,cncutoff", [(-1, 0.01, 1.5), (0, 0.2, 3.0), (100000, 0.49, 10.0),
                                                      (50, -1.0, 100.0), (400, 0.6, 1.5)])
def test_cutoff_corner_cases(ocutoff, pcutoff, cncutoff):
    g = make_inputs(2000, 7, p_chimeric=0.1, p_bubble=0.1)
    run_pipeline(g, pcutoff=pcutoff, cncutoff=cncutoff, ocutoff=ocutoff)


def test_tie_heavy_walks():
    g = make_inputs(3000, 21, dist_range_small=True, contig_median=300)
    run_pipeline(g)


def test_hub_vertices_take_the_wave_path():
    g = make_inputs(4000, 11, p_repeat=0.02, repeat_degree=300, p_chimeric=0.05)
    eng, _ = run_pipeline(g, hub_degree=8)
    assert eng.stat("hubs") > 100


def test_walk_queue_retry():
    g = make_inputs(20000, 5)
    # a walk that overflows its ring takes a larger one from the pool; with a
    # pool too small for that the whole call runs again with more room
    eng, _ = run_pipeline(g, walk_queue_factor=1, fast_walks=0, walk_pool_entries=4096)
    assert eng.stat("walk_retries") >= 1
    eng1, _ = run_pipeline(g, walk_queue_factor=1, fast_walks=0)
    assert eng1.stat("walk_retries") == 0 and eng1.digest() == eng.digest()


def test_out_of_memory_is_an_error_of_that_call_only():
    """a workspace the device cannot hold (here: a walk-queue pool of 2^40 entries)
    fails the call with GTSG_ENOMEM, leaves the states as they were, and does not
    leave a HIP error behind for the next engine of the process"""
    g = make_inputs(2000, 4)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g, walk_pool_entries=1 << 40)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(0.01, 1.5, 400); eng.filter(0.01, 1.5, 400)
    with pytest.raises(pkg.engine.EngineError, match="out of memory"):
        eng.makescaffold()
    assert_same_states(eng, og, "after the failed call")
    eng.set_option("walk_pool_entries", 1 << 20)
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold")
    run_pipeline(make_inputs(500, 9))        # a fresh engine of the same process


@pytest.mark.parametrize("seed", range(2))
def test_reference_search_for_every_walk(seed):
    g = make_inputs(5000, 60 + seed, p_chimeric=0.05)
    eng, _ = run_pipeline(g, fast_walks=0)
    assert eng.stat("fast_walks") == 0 and eng.stat("slow_walks") > 0


def test_global_memory_components_match_lds_components():
    g = make_inputs(30000, 8, p_chimeric=0.02)
    eng, _ = run_pipeline(g, lds_components=0)
    assert eng.stat("components_global_mem") == eng.stat("components")
    eng2, _ = run_pipeline(g)
    assert eng2.stat("components_global_mem") < eng2.stat("components")
    assert eng.digest() == eng2.digest()


@pytest.mark.parametrize("seed", range(3))
def test_misjoin_heavy_graphs_need_no_reference_search(seed):
    g = make_inputs(6000, 900 + seed, p_chimeric=0.15, p_inversion=0.0, p_bubble=0.05,
                    links_per_side=4, p_relist=0.05)
    eng, _ = run_pipeline(g)
    assert eng.stat("slow_walks") == 0 and eng.stat("fast_walks") > 0


def test_contig_ids_above_2_to_24():
    """more than 2^24 contigs: the pair sort's last digit reaches bit 63 of the
    key, which carries a flag that must not take part in the order"""
    g = make_inputs(6000, 41, p_relist=0.05, p_chimeric=0.03)
    off = (1 << 24) + 1234
    n = off + g["seq_len"].size
    big = dict(g)
    for k, fill in (("seq_len", 1000), ("astat", 50.0), ("copy_num", 1.0)):
        a = np.full(n, fill, dtype=g[k].dtype)
        a[off:] = g[k]
        big[k] = a
    # every second contig keeps its small id: pairs straddle the 2^24 line
    keep = np.arange(g["seq_len"].size) % 2 == 0
    newid = np.where(keep, np.arange(g["seq_len"].size), np.arange(g["seq_len"].size) + off).astype(np.uint32)
    for k in ("seq_len", "astat", "copy_num"):
        big[k][newid] = g[k]
    big["root"] = newid[g["root"]]
    big["ctg"] = newid[g["ctg"]]
    eng, og = run_pipeline(big)
    assert eng.ne == og.ne and eng.ne > 0


@pytest.mark.parametrize("scale", [1, 40, 400])
def test_distance_widths_of_the_packed_lds_layout(scale):
    """the packed LDS layout keeps the distances of a component as int16 when
    they all fit and as int32 otherwise (GtsCompMemT::cdist16); scale 40 puts
    some components above 2^15, scale 400 most of them (a few above 2^19, which
    run from global memory); the option that turns int16 off gives the same states"""
    g = make_inputs(6000, 78, p_chimeric=0.03)
    g["dist"] = g["dist"] * scale
    eng, _ = run_pipeline(g, ocutoff=400 * scale)
    eng0, _ = run_pipeline(g, ocutoff=400 * scale, lds_int16_distances=0)
    assert eng.digest() == eng0.digest()
    assert eng.stat("components_global_mem") < eng.stat("components")


@pytest.mark.parametrize("what", ["distances", "lengths", "length_sums"])
def test_values_the_packed_lds_layout_cannot_carry(what):
    """components with a distance of 2^19 or more, a contig of 2^31 bases or
    contigs adding up to 2^32 bases run from global memory (64-bit labels)"""
    g = make_inputs(4000, 77, p_chimeric=0.03)
    if what == "distances":
        g["dist"] = g["dist"] * 4096
    elif what == "lengths":
        g["seq_len"] = g["seq_len"].copy()
        g["seq_len"][::97] += 1 << 31
    else:
        g["seq_len"] = g["seq_len"] + (1 << 29)
    eng, _ = run_pipeline(g, ocutoff=400 if what == "distances" else 1 << 40)
    assert eng.stat("components_global_mem") > 0
    # the same with the walks of those components deferred: tasks that run from
    # global memory on scratch slabs of their own (k_walk_tasks_global), one
    # slab at a time and all at once
    for pool_mb in (0, 64):
        eng2, _ = run_pipeline(g, ocutoff=400 if what == "distances" else 1 << 40, defer_min_contigs=3,
                               defer_min_work=0, defer_global_components=1, global_task_pool_mb=pool_mb)
        assert eng2.stat("walk_tasks") > 0 and eng2.digest() == eng.digest()


@pytest.mark.parametrize("seed", range(3))
def test_deferred_walk_tasks(seed):
    """large clean components fan their walks out to one workgroup per
    terminal; same scaffolds as walking in place"""
    kw = dict(p_chimeric=0.08, p_inversion=0.0, p_bubble=0.05, links_per_side=4,
              unique_pairs=True) if seed else dict(p_chimeric=0.03)
    g = make_inputs(8000, 1200 + seed, **kw)
    eng, _ = run_pipeline(g, defer_min_contigs=3, defer_min_work=0)
    assert eng.stat("walk_tasks") > 0 and eng.stat("walk_task_rounds") >= 1
    eng0, _ = run_pipeline(g, defer_min_contigs=0)
    assert eng0.stat("walk_tasks") == 0
    assert eng.digest() == eng0.digest()
    # a pool too small for any component: every component walks in place
    eng1, _ = run_pipeline(g, defer_min_contigs=3, defer_min_work=0, walk_path_entries=1)
    assert eng1.digest() == eng0.digest()


def test_component_pool_matches_the_class_launches():
    """the LDS components in one launch of wavefront pools (default) or in a
    launch per size class: same states; the pool's waits never ran into their
    bound"""
    for kw in (dict(n=30000, seed=8, p_chimeric=0.02),
               dict(n=8000, seed=1201, p_chimeric=0.08, p_inversion=0.0, p_bubble=0.05, links_per_side=4,
                    unique_pairs=True)):
        kw = dict(kw)
        g = make_inputs(kw.pop("n"), kw.pop("seed"), **kw)
        eng, _ = run_pipeline(g)
        assert eng.stat("pool_gave_up_lock") <= 0 and eng.stat("pool_gave_up_pages") <= 0
        eng0, _ = run_pipeline(g, pool_components=0)
        eng1, _ = run_pipeline(g, pool_waves=3)
        assert eng.digest() == eng0.digest() == eng1.digest()
        assert sum(eng.stat("components_lds_class%d" % k) for k in range(11)) == \
            sum(eng0.stat("components_lds_class%d" % k) for k in range(11)) > 0


@pytest.mark.parametrize("poison", [0x00, 0xFF, 0xA5])
def test_programs_do_not_depend_on_stale_lds(poison):
    """a wavefront of the pool runs one component after the other in recycled
    LDS pages; with the pages overwritten by a byte pattern before staging the
    states are still the oracle's, whatever the pattern (linear walks, cyclic
    state graphs, reference searches, cycle removal, hubs)"""
    cases = [dict(n=3000, seed=21, dist_range_small=True, contig_median=300),
             dict(n=5000, seed=60, p_chimeric=0.05),
             dict(n=2000, seed=7, p_chimeric=0.1, p_bubble=0.1),
             dict(n=4000, seed=11, p_repeat=0.02, repeat_degree=300, p_chimeric=0.05),
             dict(n=6000, seed=900, p_chimeric=0.15, p_inversion=0.0, p_bubble=0.05, links_per_side=4,
                  p_relist=0.05)]
    for kw in cases:
        kw = dict(kw)
        g = make_inputs(kw.pop("n"), kw.pop("seed"), **kw)
        run_pipeline(g, lds_poison=poison)
    g = make_inputs(5000, 61, p_chimeric=0.05)
    run_pipeline(g, lds_poison=poison, fast_walks=0)


def test_fast_walks_resolve_ties():
    g = make_inputs(3000, 21, dist_range_small=True, contig_median=300)
    eng, _ = run_pipeline(g)
    assert eng.stat("fast_walks") > 0


def test_empty_and_ragged_inputs():
    # contigs without any record; records only between two contigs
    g = make_inputs(50, 1)
    for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags"):
        g[k] = g[k][:0]
    eng = engine_from_inputs(g)
    assert eng.ne == 0
    eng.mark_repeats(); eng.filter(); eng.removecycles(); eng.makescaffold()
    og = oracle_from_inputs(g); og.mark_repeats(); og.filter(); og.removecycles(); og.makescaffold(True)
    assert_same_states(eng, og, "no edges")


def test_reference_test_data(golden_dir):
    # ref testdata/primary-contigs.fa + libPE.de + libPE.astat (BASELINE configs[0])
    og = OracleGraph.from_files(golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de")
    # records re-derived from the oracle's parse: each edge pair once, in id order
    e, v = og.edges(), og.vertices()
    og.mark_repeats_file(golden_dir + "/libPE.astat")
    v2 = og.vertices()
    eng = pkg.engine.Engine(0)
    eng.set_contigs(v["seq_len"].astype(np.int64), v["astat"], v["copy_num"])
    sel = np.arange(0, og.ne, 2)
    eng.build_from_records(e["start"][sel], e["end"][sel], e["dist"][sel], e["std_dev"][sel],
                           e["num_pairs"][sel].astype(np.int64), e["flags"][sel])
    assert_same_graph(eng, og)
    eng.set_astat(v2["astat"], v2["copy_num"])
    eng.mark_repeats(True, DEFAULTS["copy_num_cutoff"], DEFAULTS["astat_cutoff"])
    assert_same_states(eng, og, "mark_repeats")
    og.filter(); eng.filter(); assert_same_states(eng, og, "filter")
    og.removecycles(); eng.removecycles(); assert_same_states(eng, og, "removecycles")
    og.makescaffold(False); eng.makescaffold(); assert_same_states(eng, og, "makescaffold")


def test_medium_graph_and_digest():
    g = make_inputs(100000, 5)
    eng, og = run_pipeline(g)
    assert eng.digest() == pkg.engine.state_digest_host(og.vertex_states(), og.edge_states())
    assert eng.stat("components") > 1000


def test_file_api_reproduces_reference_dot_files(tmp_path, golden_dir):
    """BASELINE configs[0]: testdata/primary-contigs.fa + libPE.de + libPE.astat
    through the drop-in C API, bit-exact .dot against the reference's
    *_expected.dot (ref testsuite/scaffolder_include.rb:88-121, src/test.c:130-157)."""
    import filecmp
    G = pkg.engine.ScaffolderGraph.from_files(golden_dir + "/primary-contigs.fa",
                                              golden_dir + "/libPE.de", DEFAULTS["min_ctg_len"])
    G.mark_repeats(golden_dir + "/libPE.astat", DEFAULTS["copy_num_cutoff"], DEFAULTS["astat_cutoff"])
    stages = [("mark_repeats", None), ("filter", G.filter), ("removecycles", G.removecycles),
              ("makescaffold", G.makescaffold)]
    for name, fn in stages:
        if fn:
            fn()
        out = str(tmp_path / (name + ".dot"))
        G.print_dot(out)
        assert filecmp.cmp(out, "%s/gt_scaffolder_algorithms_test_%s_expected.dot" % (golden_dir, name),
                           shallow=False), name
    # .scaf against the oracle's writer
    og = OracleGraph.from_files(golden_dir + "/primary-contigs.fa", golden_dir + "/libPE.de")
    og.mark_repeats_file(golden_dir + "/libPE.astat"); og.filter(); og.removecycles(); og.makescaffold(False)
    og.write_scaffold(str(tmp_path / "oracle.scaf"))
    G.write_scaffold(str(tmp_path / "engine.scaf"))
    assert filecmp.cmp(tmp_path / "oracle.scaf", tmp_path / "engine.scaf", shallow=False)


def test_file_api_on_synthetic_files(tmp_path):
    """BASELINE configs[1] in miniature: synthetic .fa/.de/.astat written in the
    reference's formats, parsed by both sides, identical .dot and .scaf."""
    import filecmp
    g = make_inputs(3000, 17, repeat_degree=12)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de, astat = [str(tmp_path / ("syn" + x)) for x in (".fa", ".de", ".astat")]
    og = OracleGraph.from_files(fa, de)
    G = pkg.engine.ScaffolderGraph.from_files(fa, de)
    assert (G.nv, G.ne) == (og.nv, og.ne)
    og.mark_repeats_file(astat); G.mark_repeats(astat)
    og.filter(); G.filter(); og.makescaffold(True); G.makescaffold()
    og.print_dot(str(tmp_path / "o.dot")); G.print_dot(str(tmp_path / "e.dot"))
    assert filecmp.cmp(tmp_path / "o.dot", tmp_path / "e.dot", shallow=False)
    # the edge lines are formatted on the GPU (gtsg_format_dot_edges); the host's loop gives the same file
    pkg.engine.lib().gt_scaffolder_set_dot_writer(1)
    try:
        G.print_dot(str(tmp_path / "h.dot"))
    finally:
        pkg.engine.lib().gt_scaffolder_set_dot_writer(0)
    assert filecmp.cmp(tmp_path / "o.dot", tmp_path / "h.dot", shallow=False)
    og.write_scaffold(str(tmp_path / "o.scaf"))
    lens = G.write_scaffold(str(tmp_path / "e.scaf"))
    assert filecmp.cmp(tmp_path / "o.scaf", tmp_path / "e.scaf", shallow=False)
    assert np.array_equal(lens, og.scaffolds()[3])
    # the records were ranked on the GPU (gtsg_scaffold_records); the host walk gives the same file
    assert pkg.engine.lib().gt_scaffolder_last_record_walk() == 0
    pkg.engine.lib().gt_scaffolder_set_record_walk(1)
    try:
        lens = G.write_scaffold(str(tmp_path / "h.scaf"))
    finally:
        pkg.engine.lib().gt_scaffolder_set_record_walk(0)
    assert pkg.engine.lib().gt_scaffolder_last_record_walk() == 1
    assert filecmp.cmp(tmp_path / "o.scaf", tmp_path / "h.scaf", shallow=False)
    assert np.array_equal(lens, og.scaffolds()[3])


def test_file_api_at_100k_contigs(tmp_path):
    """BASELINE configs[1] as stated: a synthetic 100 k-contig / ~1 M-edge graph
    as .fa / .de / .astat FILES in the reference's formats, parsed by the oracle
    and by the drop-in C API; .dot after every stage and .scaf byte-identical."""
    import filecmp
    g = make_inputs(100000, 5, contig_median=320, links_per_side=5, p_repeat=0.03, repeat_degree=43,
                    p_inversion=0.0, unique_pairs=True)
    pkg.synth.write_files(g, str(tmp_path / "syn"))
    fa, de, astat = [str(tmp_path / ("syn" + x)) for x in (".fa", ".de", ".astat")]
    og = OracleGraph.from_files(fa, de)
    G = pkg.engine.ScaffolderGraph.from_files(fa, de)
    assert (G.nv, G.ne) == (og.nv, og.ne) and og.nv > 60000 and og.ne > 500000
    og.mark_repeats_file(astat); G.mark_repeats(astat)
    for name, fo, fe in (("mark_repeats", None, None), ("filter", og.filter, G.filter),
                         ("removecycles", og.removecycles, G.removecycles),
                         ("makescaffold", lambda: og.makescaffold(True), G.makescaffold)):
        if fo:
            fo(); fe()
        og.print_dot(str(tmp_path / "o.dot")); G.print_dot(str(tmp_path / "e.dot"))
        assert filecmp.cmp(tmp_path / "o.dot", tmp_path / "e.dot", shallow=False), name
    og.write_scaffold(str(tmp_path / "o.scaf"))
    G.write_scaffold(str(tmp_path / "e.scaf"))
    assert pkg.engine.lib().gt_scaffolder_last_record_walk() == 0    # ranked on the GPU
    assert filecmp.cmp(tmp_path / "o.scaf", tmp_path / "e.scaf", shallow=False)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_matches_unsharded_oracle(world):
    """The multi-GPU path rehearsed on one GPU: `world` engines, records split by
    file chunk, component partition + routing, filter with the latest-hit
    exchange; merged states equal the oracle's on the whole graph."""
    import threading
    import torch
    dist_mod = pkg.dist
    g = make_inputs(6000, 41, p_repeat=0.05, p_chimeric=0.05)
    n, m = len(g["seq_len"]), len(g["root"])
    og = oracle_from_inputs(g)
    og.mark_repeats(); og.filter(); og.makescaffold(True)
    oe = og.edges(); ostate = og.edge_states()
    want = {(int(a), int(b)): int(s) for a, b, s in zip(oe["start"], oe["end"], ostate)}
    cuts = dict(copy_num_cutoff=0.3, astat_cutoff=20.0, pcutoff=0.01, cncutoff=1.5, ocutoff=400)
    shared = dist_mod.ThreadComm.Shared(world)
    res, errs = [None] * world, []

    def run(r):
        try:
            dev = "cuda:0"
            eng = pkg.engine.Engine(0)
            contigs = dict(seq_len=torch.from_numpy(g["seq_len"].astype(np.int64)).to(dev),
                           astat=torch.from_numpy(g["astat"]).to(dev),
                           copy_num=torch.from_numpy(g["copy_num"]).to(dev))
            lo, hi = m * r // world, m * (r + 1) // world
            rec = {k: torch.from_numpy(np.ascontiguousarray(g[k][lo:hi]).astype(
                {"root": np.int64, "ctg": np.int64, "num_pairs": np.int64}.get(k, g[k].dtype))).to(dev)
                for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
            rec["k"] = torch.arange(lo, hi, dtype=torch.int64, device=dev)
            owner, rounds, load, local = dist_mod.scaffold_sharded(dist_mod.ThreadComm(shared, r), eng,
                                                                   contigs, rec, cuts)
            assert eng.nv == local.numel() < n          # owned + repeat contigs only
            res[r] = (owner.cpu().numpy(), eng.vertex_states(), eng.edges(), eng.edge_states(),
                      local.cpu().numpy())
            eng.close()
        except BaseException as ex:   # noqa: B902
            errs.append(ex)
            shared.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    owner = res[0][0]
    vs = np.zeros(n, np.uint8)
    seen = {}
    for r in range(world):
        _, v, e, es, local = res[r]
        assert np.array_equal(local, np.nonzero((owner == r) | (owner < 0))[0])
        vs[local] = v                                    # repeat contigs: the same state on every rank
        for a, b, s in zip(local[e["start"]], local[e["end"]], es):
            assert (int(a), int(b)) not in seen          # every edge lives on one rank
            seen[(int(a), int(b))] = int(s)
    assert np.array_equal(vs, og.vertex_states())
    assert seen == want


def test_self_loop_and_repeated_records():
    """records whose two contigs coincide (ref parser.c:359-378 creates two edges
    r -> r) and pairs listed many times with growing / equal / shrinking std_dev"""
    g = make_inputs(1500, 77, p_chimeric=0.05)
    rng = np.random.default_rng(5)
    m = len(g["root"])
    extra = 200
    idx = rng.integers(0, m, extra)
    g2 = {k: v.copy() for k, v in g.items()}
    add = {k: g[k][idx].copy() for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
    add["ctg"][:60] = add["root"][:60]                       # self loops
    add["std_dev"][60:] = np.float32(rng.choice([0.5, 1.0, 50.0], extra - 60))   # ties and overrides
    add["flags"][120:] = rng.integers(0, 4, extra - 120).astype(np.uint8)       # other geometry
    pos = np.sort(rng.integers(0, m, extra))
    for k in add:
        g2[k] = np.insert(g[k], pos, add[k])
    run_pipeline(g2)


def test_ragged_inputs():
    """most contigs without any record, one contig linked to everything"""
    g = make_inputs(400, 9)
    n = len(g["seq_len"])
    hub = 7
    others = np.array([v for v in range(n) if v != hub][:150], dtype=np.uint32)
    k = len(others)
    rng = np.random.default_rng(1)
    rec = dict(root=np.full(k, hub, np.uint32), ctg=others,
               dist=rng.integers(-90, 3000, k).astype(np.int64),
               std_dev=(rng.random(k) * 20).astype(np.float32),
               num_pairs=rng.integers(1, 50, k).astype(np.uint64),
               flags=rng.integers(0, 4, k).astype(np.uint8))
    for name in rec:
        g[name] = rec[name]
    g["astat"][hub] = 100.0; g["copy_num"][hub] = 1.0           # the hub is not a repeat
    run_pipeline(g, hub_degree=16)


def test_full_size_properties():
    """BASELINE configs[1] (10 M contigs / 100 M edges, the workload bench.py
    times) through properties that need no oracle: structural invariants of the
    built graph, the same digest whatever the parallel decomposition (wavefront
    pool or a launch per LDS class, deferred walks or not), and the oracle's
    digest on a sample drawn by the same generator."""
    import torch
    import bench
    n = 10_000_000
    g = bench.make_inputs(pkg, n, 1234, "cuda:0", bench.WORKLOAD["gen"])
    g["num_pairs"] = g["num_pairs"].to(torch.int64)
    digests = []
    # pool launch; launch per LDS class on one stream; walks fanned out; the pool
    # with 5 wavefronts per CU and its pages overwritten before every component
    for opts in (dict(), dict(defer_min_contigs=0, pool_components=0, class_streams=1),
                 dict(defer_min_contigs=96, defer_min_work=0, mixed_task_limit=0),
                 dict(pool_waves=5, lds_poison=0xA5)):
        eng = pkg.engine.Engine(0)
        for k, v in opts.items():
            eng.set_option(k, v)
        bench.run_step(eng, g)
        digests.append(eng.digest())
        if not opts:
            assert eng.ne > 90_000_000 and eng.stat("components") > 100_000
            e = eng.edges()
            m = eng.ne
            # edge 2j and 2j+1 are created together as twins (parser.c:369-377)
            assert np.array_equal(e["start"][0:m:2], e["end"][1:m:2])
            assert np.array_equal(e["end"][0:m:2], e["start"][1:m:2])
            assert np.array_equal(e["flags"][0:m:2] & 2, e["flags"][1:m:2] & 2)   # same
            assert int(e["start"].max()) < n and int(e["end"].max()) < n
            # every contig pair owns one edge pair
            lo = np.minimum(e["start"][0:m:2], e["end"][0:m:2]).astype(np.uint64)
            hi = np.maximum(e["start"][0:m:2], e["end"][0:m:2]).astype(np.uint64)
            key = lo * np.uint64(n) + hi
            assert np.unique(key).size == key.size
            del e, lo, hi, key
            # states: an unmarked edge has unmarked ends; SCAFFOLD edges join SCAFFOLD contigs
            vs, es = eng.vertex_states(), eng.edge_states()
            ee = eng.edges()
            marked_v = np.zeros(8, bool); marked_v[[1, 3, 7]] = True
            marked_e = np.zeros(8, bool); marked_e[[1, 2, 3, 7]] = True
            assert not (~marked_e[es] & (marked_v[vs[ee["start"]]] | marked_v[vs[ee["end"]]])).any()
            sc = es == 6
            assert sc.any() and (vs[ee["start"][sc]] == 6).all() and (vs[ee["end"][sc]] == 6).all()
            assert np.array_equal(es[0:m:2] == 6, es[1:m:2] == 6)   # twins are marked together
            del vs, es, ee
        del eng
    assert digests[0] == digests[1] == digests[2] == digests[3]
    # the slow path at full size: 10 % of the false links are inversions and
    # repeated pairs are kept, so components holding them leave the linear walks
    # for the reference's label-correcting search (create_walk_reference) -- in
    # the component programs, in the walk tasks and in the select pass.  Two
    # decompositions, one digest.
    del g
    torch.cuda.empty_cache()
    gen = dict(bench.WORKLOAD["gen"], p_inversion=0.1, unique_pairs=False)
    g = bench.make_inputs(pkg, n, 1234, "cuda:0", gen)
    g["num_pairs"] = g["num_pairs"].to(torch.int64)
    slow = []
    # (the last one: no walk leaves its component -- nothing is decided by the
    # select pass and its test for walks that a revived arc made stale)
    for opts in (dict(), dict(defer_min_contigs=0, pool_components=0, class_streams=1),
                 dict(defer_min_contigs=0, defer_ref_min_contigs=0)):
        eng = pkg.engine.Engine(0)
        for k, v in opts.items():
            eng.set_option(k, v)
        bench.run_step(eng, g)
        slow.append((eng.digest(), eng.stat("slow_walks"), eng.stat("walk_tasks")))
        del eng
    assert slow[0][0] == slow[1][0] == slow[2][0] and slow[0][1] > 0 and slow[1][1] > 0, slow
    assert slow[0][2] > 0 and slow[2][2] == 0, slow
    del g
    # the oracle on a sample of the same generator
    gs = make_inputs(100000, 99, **bench.WORKLOAD["gen"])
    eng, og = run_pipeline(gs, pcutoff=bench.CUTS["pcutoff"], cncutoff=bench.CUTS["cncutoff"],
                           ocutoff=bench.CUTS["ocutoff"])
    assert eng.digest() == pkg.engine.state_digest_host(og.vertex_states(), og.edge_states())


_FULL = {}


def full_size_inputs():
    """the portable 10 M-contig workload of tests/golden/full_size_digest.json,
    drawn once per test session (53 s on a host core)"""
    import json
    import os
    import sys
    import bench
    from helpers import ROOT
    if not _FULL:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_full_size_digest as tool
        want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digest.json")))
        assert want["gen"] == bench.WORKLOADS["10M"]["gen"] and want["cuts"] == bench.CUTS
        g = tool.generate(pkg, want["n_contigs"], want["seed"], want["gen"])
        assert len(g["root"]) == want["n_records"]
        assert tool.input_checksum(g) == want["input_sha256"], "this host draws other inputs than the build container"
        _FULL.update(want=want, g=g)
    return _FULL["want"], _FULL["g"]


def test_full_size_sharded_pipeline_against_oracle_digest():
    """BASELINE configs[3] at full size, rehearsed on one GPU: the 10 M-contig /
    100 M-edge graph of the oracle fixture split over TWO engines (threads) by
    file chunk -- label, plan and route on the device, the filter's latest-hit
    exchange, makescaffold per shard -- and the merged vertex / edge states
    digested like the unsharded graph's: the oracle's digests of
    tests/golden/full_size_digest.json."""
    import threading
    import torch
    dist_mod = pkg.dist
    want, g = full_size_inputs()
    n, m = len(g["seq_len"]), len(g["root"])
    world, dev = 2, "cuda:0"
    # edge ids of the whole graph: (start, end) in id order from an unsharded build
    eng0 = pkg.engine.Engine(0)
    eng0.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    eng0.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64),
                            g["flags"])
    e0 = eng0.edges()
    eng0.close()
    ne = len(e0["start"])
    assert ne == want["n_edges"]
    key = torch.from_numpy(e0["start"].astype(np.int64)).to(dev) * n + torch.from_numpy(e0["end"].astype(np.int64)).to(dev)
    del e0
    skey, perm = torch.sort(key)
    del key
    cuts = want["cuts"]
    contigs = dict(seq_len=torch.from_numpy(g["seq_len"].astype(np.int64)).to(dev),
                   astat=torch.from_numpy(g["astat"]).to(dev), copy_num=torch.from_numpy(g["copy_num"]).to(dev))
    shared = dist_mod.ThreadComm.Shared(world)
    vs = torch.zeros(n, dtype=torch.uint8, device=dev)
    es = torch.full((ne,), 255, dtype=torch.uint8, device=dev)
    errs, lock, info, reps = [], threading.Lock(), {}, {}

    def run(r):
        try:
            eng = pkg.engine.Engine(0)
            lo, hi = m * r // world, m * (r + 1) // world
            rec = {k: torch.from_numpy(np.ascontiguousarray(g[k][lo:hi]).astype(
                {"root": np.int32, "ctg": np.int32, "num_pairs": np.int64}.get(k, g[k].dtype))).to(dev)
                for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
            rec["k"] = torch.arange(lo, hi, dtype=torch.int64, device=dev)
            owner, rounds, load, local = dist_mod.scaffold_sharded(dist_mod.ThreadComm(shared, r), eng,
                                                                   contigs, rec, cuts)
            v = torch.from_numpy(eng.vertex_states()).to(dev)
            e = eng.edges()
            st = torch.from_numpy(eng.edge_states()).to(dev)
            a = local[torch.from_numpy(e["start"].astype(np.int64)).to(dev)]
            b = local[torch.from_numpy(e["end"].astype(np.int64)).to(dev)]
            pos = torch.searchsorted(skey, a * n + b)
            with lock:
                assert bool((skey[pos] == a * n + b).all())      # every shard edge is an edge of the graph
                eid = perm[pos]
                assert bool((es[eid] == 255).all())              # ... that lives on one rank only
                es[eid] = st
                own = owner[local] == r
                vs[local[own]] = v[own]
                rep = owner[local] < 0                           # repeat contigs: on every rank, compared below
                reps[r] = (local[rep], v[rep])
                info[r] = (rounds, load.cpu().tolist(), int(local.numel()), len(e["start"]))
            eng.close()
        except BaseException as ex:   # noqa: B902
            errs.append(ex)
            shared.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    vs[reps[0][0]] = reps[0][1]
    for r in range(1, world):                                    # repeat contigs: the same state everywhere
        assert bool((vs[reps[r][0]] == reps[r][1]).all())
    assert bool((es != 255).all())                               # every edge of the graph is on some rank
    assert info[0][3] + info[1][3] == ne
    # the plan balances the records: the heavier rank holds less than 51 %
    assert max(info[0][1]) <= 0.51 * sum(info[0][1])
    dig = pkg.engine.state_digest_host(vs.cpu().numpy(), es.cpu().numpy())
    assert dig == (want["after_makescaffold"]["vertex_digest"], want["after_makescaffold"]["edge_digest"])


def test_plan_kernels_match_the_torch_plan():
    """gtsg_plan_weights / gtsg_plan_deal (the component plan of the partition
    step without a torch op or a look at the host in between) against
    dist.plan_owners, three ranks"""
    import torch
    dist_mod = pkg.dist
    dev = "cuda:0"
    g = make_inputs(20000, 43, p_repeat=0.05, p_chimeric=0.05)
    n = len(g["seq_len"])
    eng = pkg.engine.Engine(0)
    comm = dist_mod.ThreadComm(dist_mod.ThreadComm.Shared(1), 0)
    comm.world = 3            # one process plays the only shard of a three-rank plan
    comm.all_reduce = lambda t, op: t
    skip = torch.from_numpy((g["astat"] <= 20.0) | (g["copy_num"] < 0.3)).to(dev)
    root = torch.from_numpy(g["root"].astype(np.int32)).to(dev)
    ctg = torch.from_numpy(g["ctg"].astype(np.int32)).to(dev)
    labels = torch.arange(n, dtype=torch.int32, device=dev)
    eng.label_components(n, root, ctg, skip.to(torch.uint8), labels)
    assert torch.equal(labels[labels.long()], labels)      # every label is a root (the smallest contig)
    owner_t, load_t = dist_mod.plan_owners(comm, n, labels, skip, root.to(torch.int64), ctg.to(torch.int64))
    owner_e, load_e = dist_mod.plan_owners_engine(comm, eng, labels, skip.to(torch.uint8), root, ctg)
    assert owner_e.dtype == torch.int8
    assert torch.equal(owner_e.to(torch.int64), owner_t)
    assert torch.equal(load_e, load_t)
    assert int((owner_e[~skip] < 0).sum()) == 0 and int((owner_e[skip] >= 0).sum()) == 0


def test_full_size_against_oracle_digest():
    """BASELINE configs[2] against the oracle at FULL size: the 10 M-contig /
    100 M-edge workload is regenerated on this host with the generator's
    portable mode, checked against the input checksum recorded in
    tests/golden/full_size_digest.json (written in the build container by
    tools/make_full_size_digest.py, where the CPU oracle ran on the whole graph
    for ~45 min) and the engine's state digests are compared with the oracle's
    after the filter and after makescaffold."""
    want, g = full_size_inputs()
    eng = pkg.engine.Engine(0)
    eng.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64),
                           g["flags"])
    assert eng.ne == want["n_edges"]
    C = want["cuts"]
    eng.mark_repeats(True, C["copy_num_cutoff"], C["astat_cutoff"])
    eng.filter(C["pcutoff"], C["cncutoff"], C["ocutoff"])
    assert eng.digest() == (want["after_filter"]["vertex_digest"], want["after_filter"]["edge_digest"])
    assert np.bincount(eng.vertex_states(), minlength=8).tolist() == want["after_filter"]["vertex_state_counts"]
    eng.makescaffold()
    vs, es = eng.vertex_states(), eng.edge_states()
    assert np.bincount(vs, minlength=8).tolist() == want["after_makescaffold"]["vertex_state_counts"]
    assert np.bincount(es, minlength=8).tolist() == want["after_makescaffold"]["edge_state_counts"]
    assert eng.digest() == (want["after_makescaffold"]["vertex_digest"],
                            want["after_makescaffold"]["edge_digest"])


def test_50M_contig_repeat_rich_graph():
    """BASELINE configs[4] on one GPU: 50 M contigs / ~500 M edges, repeat-rich
    (1.5 % repeats with 100 links on average, a few of which look unique and
    reach the filter's hub path and the global-memory component programs).  No
    oracle at this size (the reference's per-walk O(|V|) map alone,
    algorithms.c:648-650, is 200 MB per walk): structural invariants of the
    built graph, state invariants, the same digest from two parallel
    decompositions, and the HBM the engine holds."""
    import torch
    import bench
    W = bench.WORKLOADS["50M"]
    n = W["n_contigs"]
    g = bench.make_inputs(pkg, n, 4321, "cuda:0", W["gen"])
    g["num_pairs"] = g["num_pairs"].to(torch.int64)
    torch.cuda.empty_cache()
    eng = pkg.engine.Engine(0)
    bench.run_step(eng, g)
    m = eng.ne
    assert 450_000_000 < m < 600_000_000, m
    assert eng.stat("hubs") > 100_000                      # vertices above hub_degree
    assert eng.stat("components_global_mem") > 0           # components too large for LDS
    assert eng.stat("max_component") >= 2000
    d0 = eng.digest()
    held = eng.stat("bytes_graph") + eng.stat("bytes_workspace")
    assert eng.stat("bytes_graph") >= 34 * m and held < 250 * (1 << 30), held
    print("50M graph: %d edges, %.1f GB graph + %.1f GB workspace in HBM, %d components, largest %d"
          % (m, eng.stat("bytes_graph") / 1e9, eng.stat("bytes_workspace") / 1e9,
             eng.stat("components"), eng.stat("max_component")))
    # built graph: twins, ids in range, one edge pair per contig pair
    e = eng.edges()
    assert np.array_equal(e["start"][0:m:2], e["end"][1:m:2])
    assert np.array_equal(e["end"][0:m:2], e["start"][1:m:2])
    assert np.array_equal(e["flags"][0:m:2] & 2, e["flags"][1:m:2] & 2)
    assert int(e["start"].max()) < n and int(e["end"].max()) < n
    key = torch.from_numpy(np.minimum(e["start"][0:m:2], e["end"][0:m:2]).astype(np.int64)).cuda() * n + \
        torch.from_numpy(np.maximum(e["start"][0:m:2], e["end"][0:m:2]).astype(np.int64)).cuda()
    assert torch.unique(key).numel() == key.numel()
    del key
    # states: an unmarked edge has unmarked ends; SCAFFOLD edges join SCAFFOLD contigs, in twins
    vs, es = eng.vertex_states(), eng.edge_states()
    marked_v = np.zeros(8, bool); marked_v[[1, 3, 7]] = True
    marked_e = np.zeros(8, bool); marked_e[[1, 2, 3, 7]] = True
    assert not (~marked_e[es] & (marked_v[vs[e["start"]]] | marked_v[vs[e["end"]]])).any()
    sc = es == 6
    assert sc.any() and (vs[e["start"][sc]] == 6).all() and (vs[e["end"][sc]] == 6).all()
    assert np.array_equal(es[0:m:2] == 6, es[1:m:2] == 6)
    assert (es == 1).any() and (es == 2).any() and (es == 3).any()
    del e, vs, es, sc
    # another decomposition of the same work: walks in place, a launch per LDS
    # class on one stream instead of the wavefront pool
    eng.set_option("defer_min_contigs", 0)
    eng.set_option("pool_components", 0)
    eng.set_option("class_streams", 1)
    bench.run_step(eng, g)
    assert eng.digest() == d0


def test_pair_sort_is_stable():
    """equal keys keep file order through every pass of the one-sweep radix sort:
    200 000 records over 300 contig pairs, std_dev drawn from four values, so
    which record of a pair wins a direction (the FIRST with the largest std_dev,
    ref parser.c:362) and which creates the edges depends on the order of records
    with equal pair keys; tiles of 4096 records, so every pair spans many tiles
    and the look-back chain is exercised."""
    rng = np.random.default_rng(11)
    n, k = 1000, 200_000
    a = rng.integers(0, n, 300).astype(np.uint32); b = rng.integers(0, n, 300).astype(np.uint32)
    pick = rng.integers(0, 300, k)
    flip = rng.random(k) < 0.5
    g = dict(seq_len=np.full(n, 1000, np.uint64), astat=np.full(n, 50, np.float32),
             copy_num=np.ones(n, np.float32),
             root=np.where(flip, b[pick], a[pick]).astype(np.uint32),
             ctg=np.where(flip, a[pick], b[pick]).astype(np.uint32),
             dist=rng.integers(-90, 5000, k).astype(np.int64),
             std_dev=rng.choice(np.array([1.0, 2.5, 2.5, 7.0], np.float32), k),
             num_pairs=rng.integers(1, 100, k).astype(np.uint64),
             flags=rng.integers(0, 4, k).astype(np.uint8))
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g)
    assert_same_graph(eng, og)
    # the three ways the records of a pair are brought together: sorted by the larger
    # contig only and folded per bucket whatever its length, the same with the full
    # sort taken as soon as a thread has looked at 16 records, the full sort from the start
    for opts, fell_back in ((dict(pair_bucket_limit=1 << 20), 0), (dict(pair_bucket_limit=16), 1),
                            (dict(pair_sort_full=1), 0)):
        e2 = engine_from_inputs(g, **opts)
        assert_same_graph(e2, og)
        assert e2.stat("pair_sort_fallback") == fell_back, opts
    # and with the never-replace rule of a mate-pair library (parser.c:297, :362)
    og2 = OracleGraph.from_records(g["seq_len"], g["astat"], g["copy_num"], g["root"], g["ctg"], g["dist"],
                                   g["std_dev"], g["num_pairs"], g["flags"], ismatepair=True)
    eng2 = pkg.engine.Engine(0)
    eng2.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    eng2.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64),
                            g["flags"], ismatepair=True)
    assert_same_graph(eng2, og2)
    assert not np.array_equal(eng.edges()["dist"], eng2.edges()["dist"])


def test_contig_ids_out_of_range_are_rejected():
    g = make_inputs(500, 3)
    eng = pkg.engine.Engine(0)
    eng.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
    bad = g["ctg"].copy(); bad[len(bad) // 2] = 500
    with pytest.raises(pkg.engine.EngineError, match="out of range"):
        eng.build_from_records(g["root"], bad, g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64), g["flags"])
    lab = np.arange(500, dtype=np.uint32)
    with pytest.raises(pkg.engine.EngineError, match="out of range"):
        eng.label_components(500, g["root"], bad, np.zeros(500, np.uint8), lab)
    # the engine is still usable
    eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64), g["flags"])
    assert eng.ne > 0


def test_hub_degree_set_after_build_does_not_change_the_filter():
    g = make_inputs(4000, 11, p_repeat=0.02, repeat_degree=300, p_chimeric=0.05, p_repeat_unmarked=0.3)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g)
    eng.set_option("hub_degree", 4)        # takes effect at the next build
    og.mark_repeats(); eng.mark_repeats(); og.filter(); eng.filter()
    assert_same_states(eng, og, "filter")


def test_route_kernels_match_the_torch_packing():
    """gtsg_route_pack / _unpack against dist.pack_records + a stable sort by
    destination: same 32-byte rows in the same order, same counts; unpacking
    gives the records back (with and without local renumbering)."""
    import torch
    d = pkg.dist
    rng = np.random.default_rng(4)
    n, k, world = 5000, 300_000, 5
    dev = "cuda:0"
    rec = dict(root=torch.from_numpy(rng.integers(0, n, k).astype(np.int32)).to(dev),
               ctg=torch.from_numpy(rng.integers(0, n, k).astype(np.int32)).to(dev),
               dist=torch.from_numpy(rng.integers(-2**40, 2**40, k)).to(dev),
               std_dev=torch.from_numpy((rng.random(k) * 100).astype(np.float32)).to(dev),
               num_pairs=torch.from_numpy(rng.integers(0, 2**40, k)).to(dev),
               flags=torch.from_numpy(rng.integers(0, 4, k).astype(np.uint8)).to(dev))
    first = 123_456
    rec["k"] = torch.arange(first, first + k, dtype=torch.int64, device=dev)
    owner = torch.from_numpy(rng.integers(-1, world, n)).to(dev)        # -1: shared (repeat) contig
    eng = pkg.engine.Engine(0)
    rows, counts = eng.route_pack(rec, first, owner.to(torch.int8), world)
    a, b = rec["root"].long(), rec["ctg"].long()
    dest = torch.where(owner[a] >= 0, owner[a], torch.where(owner[b] >= 0, owner[b], torch.minimum(a, b) % world))
    order = torch.sort(dest, stable=True)[1]
    assert torch.equal(rows, d.pack_records(rec)[order])
    assert counts == torch.bincount(dest, minlength=world).tolist()
    back = eng.route_unpack(rows)
    assert back["out_of_order"] == bool((back["k"][1:] < back["k"][:-1]).any())
    for name in ("root", "ctg", "dist", "num_pairs", "flags", "k"):
        assert torch.equal(back[name].long(), rec[name][order].long()), name
    assert torch.equal(back["std_dev"].view(torch.int32), rec["std_dev"][order].view(torch.int32))
    loc_of = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev)
    back2 = eng.route_unpack(rows, loc_of)
    assert torch.equal(back2["root"].long(), loc_of[a[order]].long())
    assert torch.equal(back2["ctg"].long(), loc_of[b[order]].long())
    bad = dict(rec); bad["ctg"] = rec["ctg"].clone(); bad["ctg"][7] = n
    with pytest.raises(pkg.engine.EngineError, match="out of range"):
        eng.route_pack(bad, first, owner.to(torch.int8), world)


def test_find_edge_and_alter_edge_match_a_relisted_record():
    """ref gt_scaffolder_graph.c:174-235: alter_edge on the edge find_edge returns
    is what a later record of the same (root, contig) with a larger std_dev does
    (parser.c:357-366), so the oracle built from the records plus such records
    must give the same graph and, through the whole pipeline, the same states"""
    g = make_inputs(3000, 77, p_chimeric=0.05, unique_pairs=True)
    eng = engine_from_inputs(g)
    rng = np.random.default_rng(5)
    picks = rng.choice(len(g["root"]), 40, replace=False)
    extra = {k: [] for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags")}
    for k in picks:
        r, c = int(g["root"][k]), int(g["ctg"][k])
        eid = eng.find_edge(r, c)
        assert eid is not None
        nd, nsd, nnp = int(g["dist"][k]) + 37, float(g["std_dev"].max()) + 100.0 + k % 7, 3 + k % 5
        fl = int(rng.integers(0, 4))
        eng.alter_edge(eid, nd, nsd, nnp, fl & 1, fl >> 1 & 1)
        for name, val in zip(extra, (r, c, nd, nsd, nnp, fl)):
            extra[name].append(val)
    r0 = int(g["root"][0])
    assert eng.find_edge(r0, r0) is None     # (the generator draws no self loops)
    g2 = dict(g)
    for name in extra:
        g2[name] = np.concatenate([g[name], np.array(extra[name], dtype=g[name].dtype)])
    og = oracle_from_inputs(g2)
    assert_same_graph(eng, og)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(); eng.filter()
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold after alter_edge")


def test_pool_wait_bound_leaves_a_graph_that_can_be_scaffolded_again():
    """every wait inside k_components_pool is bounded; a wavefront that runs into
    the bound leaves without entering what it waited for, the call returns
    GTSG_EINTERNAL and the states are those before the call (as for GTSG_EWALK),
    so the same call with the default bound then gives the oracle's result.
    pool_wait_limit_us = 0 makes every contended lock a bound that was hit."""
    g = make_inputs(30000, 8, p_chimeric=0.02)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(); eng.filter()
    eng.set_option("pool_wait_limit_us", 0)
    with pytest.raises(pkg.engine.EngineError) as ei:
        eng.makescaffold()
    assert "(code -6)" in str(ei.value) and "states restored" in str(ei.value)
    assert eng.stat("pool_gave_up_lock") > 0
    assert_same_states(eng, og, "after the failed call")
    eng.set_option("pool_wait_limit_us", 10_000_000)
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold after a failed call")


def test_lds_overrun_is_reported_and_leaves_a_graph_that_can_be_scaffolded_again():
    """the canary path: a word behind a component's arrays that a
    program overwrote is counted, the call returns GTSG_EINTERNAL and the states are
    those before the call.  lds_poison = 256 overwrites the word as an overrun would."""
    g = make_inputs(20000, 9, p_chimeric=0.02)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(); eng.filter()
    eng.set_option("lds_poison", 256)
    with pytest.raises(pkg.engine.EngineError) as ei:
        eng.makescaffold()
    assert "(code -6)" in str(ei.value) and "wrote past their LDS arrays" in str(ei.value)
    assert eng.stat("pool_lds_overruns") > 0
    assert_same_states(eng, og, "after the failed call")
    eng.set_option("lds_poison", -1)
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold after a failed call")


def test_walk_error_after_marks_restores_the_states():
    """GTSG_EWALK after a partial mark: with the reference's search bounded to a few
    pops the small components finish and write their marks, the larger ones give up;
    the call reports the walk error and the graph is as before, so the same call with
    the default bound gives the oracle's result."""
    g = make_inputs(6000, 13, p_chimeric=0.05)
    og = oracle_from_inputs(g)
    eng = engine_from_inputs(g, fast_walks=0)
    og.mark_repeats(); eng.mark_repeats()
    og.filter(); eng.filter()
    before_v, before_e = eng.vertex_states().copy(), eng.edge_states().copy()
    eng.set_option("max_walk_pops", 6)
    with pytest.raises(pkg.engine.EngineError) as ei:
        eng.makescaffold()
    assert "max_walk_pops" in str(ei.value) and "states restored" in str(ei.value)
    assert np.array_equal(eng.vertex_states(), before_v) and np.array_equal(eng.edge_states(), before_e)
    eng.set_option("max_walk_pops", 1 << 32)
    og.makescaffold(True); eng.makescaffold()
    assert_same_states(eng, og, "makescaffold after a walk error")


@pytest.mark.parametrize("case", range(4))
def test_team_kernel_for_global_memory_components(case):
    """components that run from global memory get a workgroup each when there
    are few of them (k_components_team): wavefront 0 runs the program, the walks
    of a cc are swept eight to a wavefront by all of them.  Forced here for every
    component (LDS off, no limit on their number); ties and inversions fall back
    to the walks made one by one"""
    kw = [dict(n=30000, seed=8, p_chimeric=0.02),
          dict(n=8000, seed=1201, p_chimeric=0.08, p_inversion=0.0, p_bubble=0.05, links_per_side=4,
               unique_pairs=True),
          dict(n=3000, seed=21, dist_range_small=True, contig_median=300),
          dict(n=6000, seed=61, p_chimeric=0.05)][case]
    kw = dict(kw)
    g = make_inputs(kw.pop("n"), kw.pop("seed"), **kw)
    eng, _ = run_pipeline(g, lds_components=0, team_max_components=1 << 30)
    assert eng.stat("team_components") == eng.stat("components") > 0
    eng0, _ = run_pipeline(g, lds_components=0, team_components=0)
    assert eng0.stat("team_components") == 0
    assert eng.digest() == eng0.digest()


@pytest.mark.parametrize("lds_bytes", [3000, 12000, 40000])
def test_team_components_that_do_not_fit_the_workgroup_lds(lds_bytes):
    """a component whose vertex arrays do not all fit the team's LDS (18 B a contig
    against 158 KB: more than ~9000 contigs) keeps what does not fit behind generic
    pointers: the terminal search on one wavefront, the level-by-level order on global
    memory, the walks' bitmaps in the slab -- the other half of every `tl_* != GTS_NONE`
    test of k_components_team.  Forced with a cap on the kernel's LDS (team_lds_bytes):
    with 3000 bytes nothing but the small components' states fits, with 12000 the queue
    of components of up to a few hundred contigs, with 40000 most of them whole."""
    g = make_inputs(12000, 1201, p_chimeric=0.08, p_inversion=0.0, p_bubble=0.05, links_per_side=4,
                    unique_pairs=True)
    eng, _ = run_pipeline(g, lds_components=0, team_max_components=1 << 30, team_lds_bytes=lds_bytes)
    assert eng.stat("team_components") == eng.stat("components") > 0


def test_bench_verify_small():
    """bench.py --verify at a small size: the headline line with the secondary workload
    behind it, the digest compared with the oracle's on the headline graph (ADVICE r03:
    the block ran after the secondary workload had released the inputs)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--contigs", "60000", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--verify"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["verified_against_oracle"] is True
    assert out["roofline"]["bound"] == "hbm" and "traffic_is" in out["roofline"]


def test_sharded_pipeline_over_rccl_one_rank():
    """the nccl branch of TorchComm had never executed.  A fresh child
    process initialises the process group (backend nccl = RCCL, one rank) before it touches
    the GPU and runs scaffold_sharded with every collective forced: all_reduce MIN / SUM /
    MAX on int32 and the all_to_all of the packed int64 rows, against the oracle."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_child.py"), "29531"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("split", [0, 1])
def test_lean_program_with_cold_list(split):
    """the alternative launch geometry of round 4 (not the default): k_components_fast runs the
    lean program, components that need the reference's search or task deferral go through the
    cold list to workgroups of the full program; with fast_split two half-pool workgroups per CU"""
    g = make_inputs(6000, 33, p_chimeric=0.1, p_bubble=0.05, p_relist_flip=0.1)
    eng, _ = run_pipeline(g, fast_components=1, fast_split=split, cold_cus=4)
    assert eng.stat("fast_kernel") == 1 and eng.stat("fast_components_done") > 0


def walk_open_part(rec, seq_len):
    """The open part of gtsg_scaffold_records walked in the reference's order of visits
    (ref gt_scaffolder_algorithms.c:925-995), as the host layer does it."""
    start, end, fl = rec["open_start"], rec["open_end"], rec["open_flags"]
    lo_of = lambda v: int(np.searchsorted(start, v, "left"))
    hi_of = lambda v: int(np.searchsorted(start, v, "right"))
    visited, out = set(), []
    for v in rec["open_root"].tolist():
        lo, hi = lo_of(v), hi_of(v)
        if v in visited or hi - lo > 1:
            continue
        visited.add(v)
        took, length = [], int(seq_len[v])
        if hi - lo == 1:
            frm, k = v, lo
            while True:
                w = int(end[k]); took.append(k)
                length += int(seq_len[w]) + int(rec["open_dist"][k])
                if w in visited:
                    break
                visited.add(w)
                sense, same = bool(fl[k] & 1), bool(fl[k] & 2)
                d = sense if same else not sense
                nxt = [q for q in range(lo_of(w), hi_of(w)) if bool(fl[q] & 1) == d and int(end[q]) != frm]
                if len(nxt) != 1:
                    break
                frm, k = w, nxt[0]
        out.append((v, took, length & (2 ** 64 - 1)))
    return out


def assert_same_records(rec, og, seq_len, max_open=None):
    roots, off, edges, seqlen = og.scaffolds()
    e = og.edges()
    want = {int(r): (edges[int(off[i]):int(off[i + 1])], int(seqlen[i])) for i, r in enumerate(roots)}
    got = {}
    for i, r in enumerate(rec["root"].tolist()):
        sl = slice(int(rec["off"][i]), int(rec["off"][i + 1]))
        got[r] = (rec["eid"][sl].astype(np.uint64), int(rec["seqlen"][i]))
        for k in ("end", "dist", "std_dev", "flags"):
            assert np.array_equal(rec[k][sl], e[k][rec["eid"][sl].astype(np.int64)]), k
    assert np.all(np.diff(rec["root"].astype(np.int64)) > 0) and np.all(np.diff(rec["open_root"].astype(np.int64)) > 0)
    for v, took, length in walk_open_part(rec, seq_len):
        assert v not in got
        got[v] = (rec["open_eid"][took].astype(np.uint64), length)
    assert sorted(got) == sorted(want)
    for r in want:
        assert np.array_equal(got[r][0], want[r][0]) and got[r][1] == want[r][1], r
    if max_open is not None:
        assert len(rec["open_eid"]) <= max_open * max(1, len(rec["eid"])), (len(rec["open_eid"]), len(rec["eid"]))


@pytest.mark.parametrize("case", ["small", "noisy", "inversions", "relisted", "100k", "empty"])
def test_scaffold_records_ranked_on_the_device(case):
    """ref gt_scaffolder_algorithms.c:901-997 as a list ranking of the clean SCAFFOLD paths
    (gtsg_scaffold_records) plus the open part walked in the reference's order: every record of
    the oracle's walk, root by root -- edges in walk order, summed length"""
    g = {"small": lambda: make_inputs(3000, 17, repeat_degree=12),
         "noisy": lambda: make_inputs(20000, 5, p_chimeric=0.3, p_bubble=0.1, p_repeat=0.05, links_per_side=3),
         "inversions": lambda: make_inputs(30000, 9, p_inversion=0.3),
         "relisted": lambda: make_inputs(20000, 11, p_relist=0.1, p_relist_flip=0.5),
         "100k": lambda: make_inputs(100000, 5, contig_median=320, links_per_side=5, p_repeat=0.03,
                                     repeat_degree=43, p_inversion=0.0, unique_pairs=True),
         "empty": lambda: make_inputs(500, 3)}[case]()
    if case == "empty":
        for k in ("root", "ctg", "dist", "std_dev", "num_pairs", "flags"):
            g[k] = g[k][:0]
    eng, og = run_pipeline(g)
    rec = eng.scaffold_records()
    assert_same_records(rec, og, g["seq_len"], max_open=None if case in ("relisted", "noisy") else 0.2)
    if case != "empty":
        assert len(rec["eid"]) > 0 and int(np.diff(rec["off"].astype(np.int64)).max()) > 2
    # and before any stage ran: no SCAFFOLD edges, every unmarked contig is a record
    eng2 = engine_from_inputs(g)
    og2 = oracle_from_inputs(g)
    rec2 = eng2.scaffold_records()
    assert len(rec2["eid"]) == 0 and len(rec2["open_root"]) == 0
    assert_same_records(rec2, og2, g["seq_len"])


def test_records_of_scaffold_paths_with_two_orientations(tmp_path):
    """A pair listed twice with another orientation (parser.c:357-366 alters one direction
    only) leaves SCAFFOLD paths whose two directions disagree: they go to the open part, the
    host walks them in the reference's order of visits; same .scaf as the oracle, and as the
    host walk of everything"""
    import filecmp
    g = make_inputs(3000, 77, p_chimeric=0.05, unique_pairs=True)
    rng = np.random.default_rng(5)
    picks = rng.choice(len(g["root"]), 400, replace=False)
    g2 = dict(g)
    extra = dict(root=g["root"][picks], ctg=g["ctg"][picks], dist=g["dist"][picks] + 37,
                 std_dev=(g["std_dev"].max() + 100.0 + picks % 7).astype(np.float32),
                 num_pairs=(3 + picks % 5).astype(g["num_pairs"].dtype),
                 flags=rng.integers(0, 4, len(picks)).astype(np.uint8))
    for k in extra:
        g2[k] = np.concatenate([g[k], extra[k].astype(g[k].dtype)])
    pkg.synth.write_files(g2, str(tmp_path / "syn"))
    fa, de, astat = [str(tmp_path / ("syn" + x)) for x in (".fa", ".de", ".astat")]
    og = OracleGraph.from_files(fa, de)
    G = pkg.engine.ScaffolderGraph.from_files(fa, de)
    og.mark_repeats_file(astat); G.mark_repeats(astat)
    og.filter(); G.filter(); og.makescaffold(True); G.makescaffold()
    og.write_scaffold(str(tmp_path / "o.scaf"))
    L = pkg.engine.lib()
    lens = G.write_scaffold(str(tmp_path / "e.scaf"))
    assert L.gt_scaffolder_last_record_walk() == 0
    assert filecmp.cmp(tmp_path / "o.scaf", tmp_path / "e.scaf", shallow=False)
    assert np.array_equal(lens, og.scaffolds()[3])
    L.gt_scaffolder_set_record_walk(1)
    try:
        G.write_scaffold(str(tmp_path / "h.scaf"))
    finally:
        L.gt_scaffolder_set_record_walk(0)
    assert L.gt_scaffolder_last_record_walk() == 1
    assert filecmp.cmp(tmp_path / "o.scaf", tmp_path / "h.scaf", shallow=False)
