"""When each launch of a KITTI-density frame is CALLED / ISSUED by the host against when its kernel starts on the device:
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -- python bench.py --workload kitti ...   (two CSVs)
    python scripts/launch_vs_kernel.py <kernel_trace.csv> <hip_api_trace.csv>
Per launch of the seven-launch frame (medians over the steady-state frames): the gap in front of the kernel, its duration, the
hipLaunchKernel call's start / end relative to the END of the previous kernel, and how long after the call returned the kernel
started.  A gap in front of a kernel whose call returned less than ~5-6 us before the previous kernel ended is the host's (and,
under the profiler, the profiler's: a traced hipLaunchKernel takes 7-11 us instead of ~3.5)."""
import csv, sys, collections
import numpy as np
def rows(path):
    r = csv.reader(open(path)); h = next(r)
    return [dict(zip(h, x)) for x in r if len(x) == len(h) and x != h]
K = rows(sys.argv[1]); A = rows(sys.argv[2])
api = {int(a["Correlation_Id"]): a for a in A if a["Function"].startswith("hipLaunchKernel") or a["Function"].startswith("hipExtLaunch")}
K.sort(key=lambda k: int(k["Start_Timestamp"]))
nm = lambda k: k["Kernel_Name"].replace("void ", "").replace("tl::", "").split("(")[0][:30]
fi = [i for i, k in enumerate(K) if nm(k).startswith("k_grid_count_all")]
fi = fi[len(fi) // 4: len(fi) * 3 // 4]
agg = collections.OrderedDict()
for a, b in zip(fi[:-1], fi[1:]):
    if b - a != 7: continue      # (the frames that need no host-added pair)
    for j, i in enumerate(range(a, b)):
        k = K[i]; c = api.get(int(k["Correlation_Id"]))
        if c is None: continue
        s, e, pe = int(k["Start_Timestamp"]), int(k["End_Timestamp"]), int(K[i - 1]["End_Timestamp"])
        agg.setdefault((j, nm(k)), []).append([(s - pe) / 1e3, (e - s) / 1e3, (int(c["Start_Timestamp"]) - pe) / 1e3,
                                               (int(c["End_Timestamp"]) - pe) / 1e3, (s - int(c["End_Timestamp"])) / 1e3])
print("launch                            gap_before    dur | call starts / returns (vs END of the previous kernel) | kernel starts after the call returned   [us, medians]")
for (j, n), v in sorted(agg.items()):
    m = np.median(np.array(v), axis=0)
    print("%d %-30s %8.2f %7.2f | %9.2f %9.2f | %7.2f   (n=%d)" % (j, n, m[0], m[1], m[2], m[3], m[4], len(v)))
