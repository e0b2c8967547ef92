"""The odometry inner loop alone (bench.py's odometry_loop block), for kernel traces.  usage: loop_probe.py [frames]"""
import sys, os, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from tloam_amd import registration as reg
args = types.SimpleNamespace(loop_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 60, seed=0)
print(bench.odometry_loop(args, reg, torch, 0))
