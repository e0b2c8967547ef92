"""One rank running the SHARDED launch forms on 1/N of the 1 M frame's source points (loop-back exchange, include/tloam_hip.h):
what a rank of an N-GPU sharded frame executes, minus the xGMI latency of the exchange.  usage: shard_loopback.py N [mailbox|mailbox_fused|rccl] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
form = sys.argv[2] if len(sys.argv) > 2 else "mailbox"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if form == "mailbox_fused":
    os.environ["TLOAM_FUSED_LARGE"] = "1"
from tloam_amd import registration as reg, synth
big = 1 << 30
cfg = reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big)
sc = synth.make_scene(seed=0, n_src=tuple(max(n // N, 16) for n in synth.M1_SRC), n_tgt=synth.M1_TGT)
H = reg.HipRegistration(cfg)
if form == "rccl":
    H.comm_init_rccl(0, 1, reg.rccl_unique_id())
else:
    H.comm_init_mailbox(0, 1, [H.comm_mailbox_export()])
H.set_frames(sc.source, sc.target)
H.gn_iter_timer(reset=True)
for _ in range(3):
    H.scan_match(sc.T_pred)
H.gn_iter_timer(reset=True)
t0 = time.perf_counter()
for _ in range(frames):
    rc, T, st = H.scan_match(sc.T_pred)
dt = (time.perf_counter() - t0) / frames
us, n = H.gn_iter_timer()
print("N %d form %s: %.4f ms/frame, GN iteration %.2f us (%d periods), sweeps %d, info %s" % (N, form, dt * 1e3, us / max(n, 1), n, st["gn_sweeps"], H.info()))
