"""Per-frame launch timeline from a rocprofv3 rocpd database (kernel-trace of `bench.py --workload kitti ...`):
for the steady-state frames, the sequence of kernels of one frame with duration and the gap to the previous
kernel's end -- where a frame's wall time goes (kernel time vs dispatch gaps vs host round trips)."""
import sqlite3, sys, collections
import numpy as np
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
names = [r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("tl::", "") for r in rows]
st = np.array([r[1] for r in rows], float); en = np.array([r[2] for r in rows], float)
fi = [i for i, n in enumerate(names) if n.startswith("k_grid_count_all") or n.startswith("k_grid_build_coop")]     # a frame starts with the grid build
fi = fi[len(fi) // 2: len(fi) // 2 + 40]
agg = collections.OrderedDict()
frames = []
for a, b in zip(fi[:-1], fi[1:]):
    frames.append((st[b] - st[a]) / 1e3)
    for k, i in enumerate(range(a, b)):
        gap = (st[i] - en[i - 1]) / 1e3
        key = (k, names[i])
        agg.setdefault(key, []).append(((en[i] - st[i]) / 1e3, gap))
print("frames %d, period us mean %.1f p50 %.1f" % (len(frames), np.mean(frames), np.median(frames)))
tk = tg = 0.0
modal = collections.Counter(len([1 for k in agg if True]) for _ in [0])
for (k, n), v in agg.items():
    if len(v) < len(frames) * 0.6:
        continue
    d = np.median([x[0] for x in v]); g = np.median([x[1] for x in v])
    tk += d; tg += g
    print("%3d %-28s dur %6.2f  gap_before %6.2f   (n=%d)" % (k, n[:28], d, g, len(v)))
print("sum of medians: kernels %.1f us, gaps %.1f us" % (tk, tg))
