"""Host memory of the process over cycles of engine create / run / destroy
(GPU box).  usage: python tools/leak_probe.py [cycles]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import psutil  # noqa: E402
from helpers import make_inputs, oracle_from_inputs, pkg  # noqa: E402
from test_gpu_parity import engine_from_inputs  # noqa: E402

proc = psutil.Process()


def rss():
    return proc.memory_info().rss / 2**20


def phase(name, fn, cycles):
    r0 = rss()
    t0 = time.time()
    for i in range(cycles):
        fn(i)
        if rss() - r0 > 20000:
            print(name, "stopped at cycle", i, "RSS +%.0f MB" % (rss() - r0), flush=True)
            return
    print("%-34s %4d cycles  RSS %+8.1f MB  (%.1f MB / cycle)  %.1f s" % (
        name, cycles, rss() - r0, (rss() - r0) / cycles, time.time() - t0), flush=True)


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    g = make_inputs(3000, 5, p_chimeric=0.05)

    def create_destroy(i):
        pkg.engine.Engine(0).close()

    def build_only(i):
        eng = engine_from_inputs(g)
        eng.close()

    def pipeline(i):
        eng = engine_from_inputs(g)
        eng.mark_repeats(); eng.filter(0.01, 1.5, 400); eng.makescaffold()
        eng.vertex_states(); eng.edge_states()
        eng.close()

    def oracle_only(i):
        og = oracle_from_inputs(g)
        og.mark_repeats(); og.filter(0.01, 1.5, 400); og.makescaffold(True)

    def inputs_only(i):
        make_inputs(3000, 100 + i, p_chimeric=0.05)

    only = sys.argv[2] if len(sys.argv) > 2 else ""
    create_destroy(0); pipeline(0)
    for name, fn in (("engine create + destroy", create_destroy), ("create + build + destroy", build_only),
                     ("create + pipeline + destroy", pipeline), ("oracle pipeline", oracle_only),
                     ("synthetic inputs", inputs_only)):
        if only in name:
            phase(name, fn, cycles)


if __name__ == "__main__":
    main()
