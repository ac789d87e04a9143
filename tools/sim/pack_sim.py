#!/usr/bin/env python3
"""Event-driven model of how the component programs pack onto the GPU.

Input: the file GTS_DUMP_COMPONENTS names (six uint64 per component: contigs,
compact edges, ticks of removecycles, other, linear walks, reference walks;
100 MHz), written by the engine in a profile-2 step, e.g.
    GTS_DUMP_COMPONENTS=/tmp/comps.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/sim/pack_sim.py /tmp/comps.bin [extra]
Each CU has 160 KB of LDS and S wave slots; a component needs its packed LDS
footprint (rounded to pages) and one slot for its measured wave time.

policies
  classes   the engine today: one launch per size class (footprint = class size),
            classes handed out largest first on `streams` streams
  twoended  one persistent launch: a free slot takes the largest waiting
            component that fits the CU's free LDS, else the smallest
"""
import heapq
import sys
import numpy as np

CUS, LDS = 256, 160 * 1024
KL = [4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 163840]


def lds_bytes(nv, ne):
    a = 16
    r = lambda x: (x + a - 1) // a * a
    return r((nv + 1) * 2) * 2 + r(nv * 2) * 11 + r(nv * 4) * 4 + r(nv) * 4 + r(ne * 2) + r(ne * 4) + r(ne)


def twoended(need, t, slots, page=1024, helper=None):
    """need: bytes, t: us; returns makespan (us)"""
    order = np.argsort(-need, kind="stable")
    need = ((need[order] + page - 1) // page * page).astype(np.int64)
    t = t[order]
    lo, hi = 0, len(need) - 1
    free = [LDS] * CUS
    used = [0] * CUS
    ev = []   # (time, cu, bytes)
    now = 0.0
    # initial fill round-robin over CUs
    def try_start(cu):
        nonlocal lo, hi
        started = False
        while used[cu] < slots and lo <= hi:
            if need[lo] <= free[cu]:
                i = lo; lo += 1
            elif need[hi] <= free[cu]:
                i = hi; hi -= 1
            else:
                break
            free[cu] -= need[i]; used[cu] += 1
            heapq.heappush(ev, (now + t[i], cu, need[i]))
            started = True
        return started
    # breadth-first initial placement: one per CU per sweep so big ones spread
    progress = True
    while progress:
        progress = False
        for cu in range(CUS):
            if used[cu] < slots and lo <= hi:
                if need[lo] <= free[cu]:
                    i = lo; lo += 1
                elif need[hi] <= free[cu]:
                    i = hi; hi -= 1
                else:
                    continue
                free[cu] -= need[i]; used[cu] += 1
                heapq.heappush(ev, (t[i], cu, need[i]))
                progress = True
    while ev:
        now, cu, b = heapq.heappop(ev)
        free[cu] += b; used[cu] -= 1
        try_start(cu)
    return now


def classes(need, t, slots, streams=6, queues=4):
    """class launches in stream order; within the running launches workgroups are
    dispatched round-robin, big classes first"""
    k = np.searchsorted(KL, need)
    nk = len(KL)
    per_stream = [[] for _ in range(streams)]
    for j, kk in enumerate(range(nk - 1, -1, -1)):
        per_stream[j % streams].append(kk)
    items = {kk: list(np.nonzero(k == kk)[0]) for kk in range(nk)}
    left = {kk: len(items[kk]) for kk in range(nk)}        # not finished
    free = [LDS] * CUS
    used = [0] * CUS
    ev = []
    now = 0.0
    active = [s.pop(0) if s else None for s in per_stream]

    def dispatch():
        # greedy: for every CU, take from active launches (largest class first)
        for cu in range(CUS):
            while used[cu] < slots:
                done = True
                for kk in sorted([a for a in active if a is not None], reverse=True):
                    if items[kk] and KL[kk] <= free[cu]:
                        i = items[kk].pop()
                        free[cu] -= KL[kk]; used[cu] += 1
                        heapq.heappush(ev, (now + t[i], cu, kk))
                        done = False
                        break
                if done:
                    break
    dispatch()
    while ev:
        now, cu, kk = heapq.heappop(ev)
        free[cu] += KL[kk]; used[cu] -= 1
        left[kk] -= 1
        if left[kk] == 0:
            for s in range(streams):
                if active[s] == kk:
                    active[s] = per_stream[s].pop(0) if per_stream[s] else None
        dispatch()
    return now


def load(path):
    """-> (contigs, edges, ticks without walks, ticks of walks) per component"""
    if path.endswith(".npy"):
        return np.load(path).astype(np.int64)
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
    return np.stack([a[:, 0], a[:, 1], a[:, 2] + a[:, 3], a[:, 4] + a[:, 5]], 1)


def main(path):
    a = load(path)
    nv, ne = a[:, 0], a[:, 1]
    t = (a[:, 2] + a[:, 3]) / 100.0
    need = np.array([lds_bytes(int(x), int(y)) for x, y in zip(nv, ne)], dtype=np.int64)
    print("components %d  wave time %.1f s  longest %.0f us  LDS*time bound %.2f ms  slot bound(20) %.2f ms"
          % (len(t), t.sum() / 1e6, t.max(), (need * t).sum() / (CUS * LDS) / 1e3, t.sum() / (CUS * 20) / 1e3))
    for slots in (16, 20, 32):
        print("twoended slots=%d: %.2f ms" % (slots, twoended(need, t, slots) / 1e3))
    for streams in (6, 11):
        print("classes slots=20 streams=%d: %.2f ms" % (streams, classes(need, t, 20, streams) / 1e3))
    # walks on two wavefronts for components above 16 KB: walk time halves, footprint + 22 B / contig
    big = need > 16384
    t2 = np.where(big, a[:, 2] / 100.0 + a[:, 3] / 200.0, t)
    need2 = np.where(big, need + 22 * nv, need)
    print("twoended slots=20, pair walks >16K: %.2f ms" % (twoended(need2, t2, 20) / 1e3))
    # walks spread over up to 8 helpers for components above 16 KB
    for h in (4, 8):
        th = np.where(big, a[:, 2] / 100.0 + a[:, 3] / 100.0 / h, t)
        needh = np.where(big, need + 22 * nv * (h - 1), need)
        print("twoended slots=20, %d walkers >16K (ideal split): %.2f ms" % (h, twoended(needh, th, 20) / 1e3))




def extra(path):
    a = load(path)
    nv, ne = a[:, 0], a[:, 1]
    t = (a[:, 2] + a[:, 3]) / 100.0
    need = np.array([lds_bytes(int(x), int(y)) for x, y in zip(nv, ne)], dtype=np.int64)
    global LDS
    LDS = 156 * 1024
    for page in (1024, 2048, 4096):
        for slots in (16, 20):
            print("pool 156K page=%d slots=%d: %.2f ms" % (page, slots, twoended(need, t, slots, page) / 1e3))
    # helpers: components with walk time > 500 us get their walks split over h waves (each a slot)
    for h in (2, 4, 8):
        big = a[:, 3] / 100.0 > 400
        # model: owner time = rc + walks/h; helpers = (h-1) extra items of walks/h time, 22 B/contig
        t_o = np.where(big, a[:, 2] / 100.0 + a[:, 3] / 100.0 / h, t)
        extra_t = np.repeat(a[big, 3] / 100.0 / h, h - 1)
        extra_need = np.repeat(22 * nv[big] + 64, h - 1)
        tt = np.concatenate([t_o, extra_t]); nn = np.concatenate([need, extra_need])
        print("pool page=2048 slots=16 helpers=%d (on %d comps): %.2f ms  longest %.0f us" %
              (h, big.sum(), twoended(nn, tt, 16, 2048) / 1e3, t_o.max()))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    (extra if sys.argv[2:3] == ["extra"] else main)(sys.argv[1])
