"""Builds tloam_amd/libtloam_hip.so (hand-written HIP kernels + the C ABI) for gfx950 with hipcc.

In-tree on purpose: the .so travels with the repo snapshot to the GPU box.  No CPU fallback is
built -- if hipcc is missing this raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtloam_hip.so")
OBJ = os.path.join(CSRC, "_obj")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# kernel arguments: the first 12 dwords of every kernel arrive preloaded in SGPRs (gfx950 command processor) instead of
# behind a scalar-load round trip -- the K3 / minimiser / finish kernels order their arguments for it (tl_gn.hip)
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=12"]
COMMON += os.environ.get("TLOAM_EXTRA_HIPCC_FLAGS", "").split()  # development aid (-DTLOAM_K3_PROFILE ...)
UNITS = [
    # K1/K2: un-fused fp64 so the discontinuous gates see the oracle's operation order
    ("tl_nn.hip", ["-ffp-contract=off", *PRELOAD]),
    ("tl_submap.hip", ["-ffp-contract=off"]),   # voxel indices / means in the oracle's operation order
    ("tl_feature.hip", ["-ffp-contract=off"]),  # PCA gates (flatness / cvr thresholds) like the oracle
    # K3 / K5: no implicit contraction -- the same inlined residual code is compiled into several kernels (streaming sweep,
    # one-wave-per-chunk sweep, the one-launch Solve) and has to round alike in all of them (the tests compare those paths
    # bit for bit; with `fast` and with `on` the optimiser fused the same source line differently from kernel to kernel):
    # every fused multiply-add in tl_gn.hip / tl_step.hpp is spelled __builtin_fma
    ("tl_gn.hip", ["-ffp-contract=off", *PRELOAD]),
    ("tl_api.hip", []),          # context lifetime, configuration, sharding rules
    ("tl_api_frames.hip", []),   # HBM residency: hand-over of the clouds, search grids, staged frames
    ("tl_api_match.hip", []),    # the scanMatching driver
    ("tl_api_comm.hip", []),     # multi-GPU exchange (RCCL at run time, callback, mailbox)
    ("tl_api_submap.hip", []),
    ("tl_api_feature.hip", []),
    ("tl_probe.hip", []),        # read-stream bandwidth probe (the on-box ceiling of the bench's roofline block)
]
HEADERS = ["tl_common.hpp", "tl_se3.hpp", "tl_knn.hpp", "tl_walk.hpp", "tl_ctx.hpp", "tl_step.hpp", "tl_finish.hpp", "tl_prep.hpp", os.path.join("..", "..", "include", "tloam_hip.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")
    return exe


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = [os.path.join(CSRC, u) for u, _ in UNITS] + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(s) > t for s in srcs)


def build_variant(name, flags, units=("tl_gn.hip",), verbose=False):
    """Tuning builds (never the product library): tloam_amd/_variants/lib_<name>.so = the shipped objects with
    `units` recompiled under extra -D flags.  Selected at run time with TLOAM_HIP_LIB=<path>."""
    build(force=False, verbose=verbose)
    hipcc = _hipcc()
    vdir = os.path.join(HERE, "_variants")
    odir = os.path.join(vdir, "obj_" + name)
    os.makedirs(odir, exist_ok=True)
    objs, procs = [], []
    for src, extra in UNITS:
        if src in units:
            obj = os.path.join(odir, src.replace(".hip", ".o"))
            if "-DNO_PRELOAD" in flags:   # A/B of the kernel-argument preload
                extra = [f for f in extra if f not in PRELOAD]
            cmd = [hipcc, *COMMON, *extra, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        else:
            obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    out = os.path.join(vdir, f"lib_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    procs = []
    for src, extra in UNITS:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m tloam_amd.build --variant NAME [--units a.hip,b.hip] -- -DFOO=1 ...
        i = sys.argv.index("--variant")
        units = ("tl_gn.hip",)
        if "--units" in sys.argv:
            units = tuple(sys.argv[sys.argv.index("--units") + 1].split(","))
        flags = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        print(build_variant(sys.argv[i + 1], flags, units, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
