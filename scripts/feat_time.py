"""PCA feature extraction (extract_planar_sphere) of a 100 k-point cloud, host call to host return; run under
rocprofv3 --kernel-trace --stats for the kernels' own times (k_pca_info)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tloam_amd import registration as reg, synth_submap as ss
H = reg.HipRegistration(reg.default_config())
cloud = ss.feature_cloud(0, n=100000)
for _ in range(3):
    H.extract_planar_sphere(cloud)
t0 = time.perf_counter()
for _ in range(10):
    lists = H.extract_planar_sphere(cloud)
print("feature_extract_ms %.4f" % ((time.perf_counter() - t0) / 10 * 1e3), [len(x) for x in lists])
