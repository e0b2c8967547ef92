"""Probe (CPU, ~1 min): would neighbour sets carried between outer GNC iterations pay on the 1 M frame?  Steps the C oracle through
the frame, and for every re-search (outer iterations 1..3) evaluates the exact carry criterion of DESIGN.md section 9 -- every
member of the old top-K closer than min(old (K+1)-th distance, search reach) - |dq| -- with scipy's k-d tree: the fraction of
queries that could skip the search, per kind.  Round 5 result: |dq| median 15 cm / 1.2 cm / 5 mm at outer iterations 1 / 2 / 3;
4-19 % / 62-81 % / 76-90 % of the planar, ground and edge queries carry (sphere: 37 %)."""
import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from scipy.spatial import cKDTree
from tloam_amd import synth
from oracle import binding as ob
BIG = 1 << 30
sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
O = ob.Oracle(ob.make_config(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG), builder_threads=8, eval_threads=8, fast=True)
O.set_frames(sc.source, sc.target)
assert O.sm_begin(sc.T_pred) == 0
xs = [ob.se3_log(sc.T_pred)]
done = False
t = time.time()
while not done:
    rc, done, st = O.sm_outer()
    xs.append(np.array(st["se3"]))
    print("outer", st["outer_iterations"], "accepted", st["accepted_steps"], "n_corr", st["n_corr"], "%.1f s" % (time.time() - t), flush=True)
radius = [0.5, 0.5, 1.0, 0.5]
for k in range(4):
    src, tgt = sc.source.cloud(k), sc.target.cloud(k)
    tree = cKDTree(tgt)
    K = 1 if k == 3 else 5
    for it in range(1, len(xs) - 1):
        # search done at pose xs[it] (build of outer iteration `it`); memo from the search at xs[it_memo]
        T1 = ob.se3_exp(xs[it])
        p1 = src @ T1[:3, :3].T + T1[:3, 3]
        for memo in ([it - 1] if it == 1 else [it - 1, 1]):
            T0 = ob.se3_exp(xs[memo])
            p0 = src @ T0[:3, :3].T + T0[:3, 3]
            dq = np.linalg.norm(p1 - p0, axis=1)
            d, i = tree.query(p0, k=K + 1)
            reach = radius[k] * (1 + 1e-6)
            B = np.minimum(d[:, K], reach) - dq - 1e-9
            dn = np.linalg.norm(tgt[i[:, :K]] - p1[:, None, :], axis=2)      # members' new distances
            ok = np.all(dn < B[:, None], axis=1)
            print("kind", k, "build at outer", it, "memo from", memo, "dq median %.2e max %.2e" % (np.median(dq), dq.max()),
                  "carry ok %.1f %%" % (100 * ok.mean()), flush=True)
