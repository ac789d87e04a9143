"""Debug aid: one synthetic graph through the pipeline with the component kernels'
statistics printed (fast / cold split).  usage: python tools/dbg_fast.py n seed [opt=value ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import make_inputs, oracle_from_inputs, pkg

n, seed = int(sys.argv[1]), int(sys.argv[2])
kw = dict(p_chimeric=0.3, p_bubble=0.1, p_repeat=0.05, links_per_side=3)
opts = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    if k.startswith("gen."):
        kw[k[4:]] = float(v) if "." in v else int(v)
    else:
        opts[k] = int(v)
g = make_inputs(n, seed, **kw)
og = oracle_from_inputs(g)
eng = pkg.engine.Engine(0)
eng.set_option("profile", 1)
for k, v in opts.items():
    eng.set_option(k, v)
eng.set_contigs(g["seq_len"].astype(np.int64), g["astat"], g["copy_num"])
eng.build_from_records(g["root"], g["ctg"], g["dist"], g["std_dev"], g["num_pairs"].astype(np.int64), g["flags"])
og.mark_repeats(); eng.mark_repeats()
og.filter(0.01, 1.5, 400); eng.filter(0.01, 1.5, 400)
keys = ["components", "max_component", "components_global_mem", "fast_kernel", "fast_components_done",
        "fast_components_handed_over", "fast_us_first_exit", "fast_us_last_exit", "cold_us_last_exit_after_fast_start",
        "pool_us_first_exit", "pool_us_last_exit", "pool_gave_up_lock", "pool_gave_up_claim", "pool_gave_up_pages",
        "pool_lds_overruns", "fast_walks", "slow_walks", "clean_components"]
for stage in ("removecycles", "makescaffold"):
    try:
        getattr(og, stage)(*((True,) if stage == "makescaffold" else ()))
        getattr(eng, stage)()
        ok = bool(np.array_equal(eng.vertex_states(), og.vertex_states()) and np.array_equal(eng.edge_states(), og.edge_states()))
        print(stage, "parity", ok)
    except Exception as ex:
        print(stage, "FAILED:", ex)
    print(json.dumps({k: eng.stat(k) for k in keys}))
