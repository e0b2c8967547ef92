"""In-kernel timeline of K3 on the 1 M pre-built set.  Needs a profiling build:
   TLOAM_EXTRA_HIPCC_FLAGS=-DTLOAM_K3_PROFILE python -m tloam_amd.build --force
Wave 0 of every block stamps s_memtime at entry, after the state load, after the sweep and after the block
reduction; printed as a distribution over blocks."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
sets, x_true, x_eval = synth.make_prebuilt(seed=1, weights="timing")
H = reg.HipRegistration()
for rt in range(3):
    p, a, b, d, w = sets[rt]
    H.set_correspondences(rt, p, a, b, d, w)
us = [H.time_accumulate(x_eval, 50) for _ in range(3)]
print("K3 us/launch", us)
reps = int(os.environ.get("K3_PROFILE_REPS", "1"))
L = H.L
allrows = []
for _ in range(reps):
    H.time_accumulate(x_eval, 1)
    b_ = np.zeros(4096 * 32)
    nb_ = L.tloam_debug_partials(H.h, b_.ctypes.data_as(C.POINTER(C.c_double)), b_.size)
    allrows.append(b_[: nb_ * 32].reshape(nb_, 32)[:, 28:32].copy())
if os.environ.get("K3_PROFILE_DUMP"):
    np.save(os.environ["K3_PROFILE_DUMP"], np.stack(allrows))
buf = np.zeros(4096 * 32)
nb = L.tloam_debug_partials(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
rows = buf[: nb * 32].reshape(nb, 32)

wc = rows[:, 28]                                # wall clock (10 ns ticks, low 32 bits) at entry
wdur = np.floor(rows[:, 31] / 65536.0)          # wall-clock ticks entry -> end of block
rows[:, 31] -= wdur * 65536.0
tick_ns = (wdur * 10.0).sum() / (rows[:, 29] + rows[:, 30] + rows[:, 31]).sum()
print("s_memtime tick = %.3f ns" % tick_ns)
t0 = (wc - wc.min()) * 10.0 / tick_ns
ld, sw, rd = rows[:, 29], rows[:, 30], rows[:, 31]
end = t0 + ld + sw + rd
def q(v): return "min %.0f p50 %.0f p90 %.0f max %.0f" % (v.min(), np.median(v), np.percentile(v, 90), v.max())
print("blocks", nb, "(ticks of s_memtime)")
print("start offset :", q(t0))
print("state load   :", q(ld))
print("sweep        :", q(sw))
print("reduce+store :", q(rd))
print("end          :", q(end), " -> span", end.max())
# who is slow?  by XCD (blocks are dealt round-robin to the 8 XCDs), by position in the grid, by start time
sw_us = sw * tick_ns / 1e3
xcd = np.arange(nb) % 8
print("sweep us by XCD  :", " ".join("%d:%.2f/%.2f" % (x, np.median(sw_us[xcd == x]), sw_us[xcd == x].max()) for x in range(8)))
dec = np.array_split(np.arange(nb), 10)
print("sweep us by grid decile (p50/max):", " ".join("%.2f/%.2f" % (np.median(sw_us[i]), sw_us[i].max()) for i in dec))
order = np.argsort(t0)
dec = np.array_split(order, 10)
print("sweep us by start-time decile    :", " ".join("%.2f/%.2f" % (np.median(sw_us[i]), sw_us[i].max()) for i in dec))
print("end us by start-time decile (max):", " ".join("%.2f" % (end[i].max() * tick_ns / 1e3) for i in dec))
slow = np.argsort(-sw_us)[:16]
print("slowest blocks:", [(int(b), round(float(sw_us[b]), 2), round(float(t0[b] * tick_ns / 1e3), 2)) for b in slow])
