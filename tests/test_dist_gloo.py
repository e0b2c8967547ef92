"""CPU, world_size 2 / 4 / 8, gloo: the multi-rank decomposition of the path.

The product shards SOURCE points in contiguous index blocks (tloam_shard_range; a whole Frame: tloam_shard_ranges_frame -- the
four clouds end to end, cut into equal pieces), replicates the targets, and sums per-rank normal equations with one all-reduce
per sweep.  Without a GPU the per-shard
arithmetic is supplied by the oracle (test infrastructure); what is under test is the decomposition:
  * shard ranges (the product's tloam_shard_range, a host function of the C ABI),
  * sum over ranks of per-shard (H, g, cost) == the unsharded sweep,
  * the index-order cap across ranks: "added iff valid and (#counted on lower ranks + local prefix) < cap"
    reproduces the single-rank correspondence list (the rule k_compact implements).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import binding as ob
from tloam_amd import registration as reg
from tloam_amd import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130, weights="timing")
        O = ob.Oracle()
        for rt in range(3):
            p, a, b, d, w = sets[rt]
            lo, hi = reg.shard_range(len(p), rank, world)
            O.set_correspondences(rt, p[lo:hi], a[lo:hi], None if b is None else b[lo:hi],
                                  None if d is None else d[lo:hi], w[lo:hi])
        H, g, cost = O.accumulate(x_eval)
        buf = torch.zeros(48, dtype=torch.float64)           # the 48-double exchange buffer of the product
        iu = np.triu_indices(6)
        buf[:21] = torch.from_numpy(H[iu]); buf[21:27] = torch.from_numpy(g); buf[27] = cost
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        # ---- cap prefix across ranks: counted totals of lower ranks
        sc = synth.make_scene(seed=6)
        cap = 60
        F = ob.Oracle(ob.make_config(planar_maxnum=1 << 30)); F.set_frames(sc.source, sc.target)
        F.sm_begin(sc.T_pred); F.sm_outer()
        valid_idx = F.get_correspondences(0)["idx"]          # all valid planar slots, index order
        n_src = len(sc.source.planar)
        lo, hi = reg.shard_range(n_src, rank, world)
        local_valid = valid_idx[(valid_idx >= lo) & (valid_idx < hi)]
        counts = torch.zeros(world, dtype=torch.float64); counts[rank] = len(local_valid)
        dist.all_reduce(counts)                               # how the product exchanges the 4 x nranks counts
        offset = int(counts[:rank].sum())
        added = [int(i) for j, i in enumerate(local_valid) if offset + j < cap]
        gathered = [None] * world
        dist.all_gather_object(gathered, added)
        # ---- a whole Frame over the ranks (tloam_shard_ranges_frame), binding caps on ALL FOUR kinds: every rank keeps the
        #      intersection of its piece of the concatenated clouds with each kind, the cap prefix runs over the lower ranks'
        #      counts per kind -- the merged lists must be the single-rank lists
        caps = dict(planar_maxnum=60, ground_maxnum=45, edge_maxnum=25, sphere_maxnum=12)
        big = {k: 1 << 30 for k in caps}
        G = ob.Oracle(ob.make_config(**big)); G.set_frames(sc.source, sc.target)
        G.sm_begin(sc.T_pred); G.sm_outer()
        n4 = [len(sc.source.cloud(k)) for k in range(4)]
        ranges = reg.shard_ranges_frame(n4, rank, world)
        capv = [caps["planar_maxnum"], caps["ground_maxnum"], caps["edge_maxnum"], caps["sphere_maxnum"]]
        cnt4 = torch.zeros(world, 4, dtype=torch.float64)
        local = []
        for k in range(4):
            # (planar / ground / edge: the cap counts ADDED factors, i.e. valid slots; the sphere builder counts every source
            #  point it looks at, :551 -- the product's `counted` flag; here: indices below the cap are the ones ever looked at)
            vk = G.get_correspondences(k)["idx"]
            lo, hi = ranges[k]
            lv = vk[(vk >= lo) & (vk < hi)]
            local.append(lv)
            cnt4[rank, k] = (hi - lo) if k == 3 else len(lv)
        dist.all_reduce(cnt4)                                 # the 4 x nranks zero-padded sum the product exchanges
        added4 = []
        for k in range(4):
            off = int(cnt4[:rank, k].sum())
            if k == 3:   # counted = every source point: position in the cloud, added iff valid and counted-before < cap
                lo, hi = ranges[k]
                added4.append([int(i) for i in local[k] if off + (int(i) - lo) < capv[k]])
            else:
                added4.append([int(i) for j, i in enumerate(local[k]) if off + j < capv[k]])
        gathered4 = [None] * world
        dist.all_gather_object(gathered4, (added4, ranges))
        if rank == 0:
            q.put((buf.numpy().copy(), sum(gathered, []), gathered4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_decomposition_equals_single_rank(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    buf, added, frame4 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    # single-rank references
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130, weights="timing")
    O = ob.Oracle()
    for rt in range(3):
        p_, a, b, d, w = sets[rt]
        O.set_correspondences(rt, p_, a, b, d, w)
    H, g, cost = O.accumulate(x_eval)
    iu = np.triu_indices(6)
    np.testing.assert_allclose(buf[:21], H[iu], rtol=1e-12, atol=1e-12 * np.abs(H).max())
    np.testing.assert_allclose(buf[21:27], g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
    assert abs(buf[27] - cost) < 1e-12 * cost
    sc = synth.make_scene(seed=6)
    S = ob.Oracle(ob.make_config(planar_maxnum=60)); S.set_frames(sc.source, sc.target)
    S.sm_begin(sc.T_pred); S.sm_outer()
    assert added == list(S.get_correspondences(0)["idx"])     # capped list == concatenation of per-rank lists
    # the whole Frame over the ranks: blocks tile every cloud in rank order, the pieces are equal, and with binding caps on all
    # four kinds the merged per-rank lists are the single-rank lists
    n4 = [len(sc.source.cloud(k)) for k in range(4)]
    for k in range(4):
        edges = [frame4[r][1][k] for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n4[k] and all(edges[r][1] == edges[r + 1][0] for r in range(world - 1)), (k, edges)
    sizes = [sum(hi - lo for lo, hi in frame4[r][1]) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == sum(n4)
    if world >= 4:
        assert max(sum(1 for lo, hi in frame4[r][1] if hi > lo) for r in range(world)) <= 2   # a rank touches at most two kinds (here)
    C4 = ob.Oracle(ob.make_config(planar_maxnum=60, ground_maxnum=45, edge_maxnum=25, sphere_maxnum=12)); C4.set_frames(sc.source, sc.target)
    C4.sm_begin(sc.T_pred); C4.sm_outer()
    for k in range(4):
        merged = sum((frame4[r][0][k] for r in range(world)), [])
        assert merged == list(C4.get_correspondences(k)["idx"]), k
