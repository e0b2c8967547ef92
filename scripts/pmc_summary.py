"""Summarise rocprofv3 --pmc passes (rocpd sqlite) for one kernel: per-launch mean of each counter over all
launches and over the 'working' launches (counter value above 50 % of the maximum, i.e. not the no-op launches
of an already-finished Solve).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; FETCH_SIZE gets the
gfx950 x2 correction for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section).

usage: pmc_summary.py <kernel-substring> <out.json> <db> [<db> ...]
"""
import json
import sqlite3
import sys


def collect(db, kernel):
    c = sqlite3.connect(db)
    rows = c.execute("select counter_name, dispatch_id, sum(value) from counters_collection "
                     "where kernel_name like ? group by counter_name, dispatch_id", (f"%{kernel}%",)).fetchall()
    out = {}
    for name, _, v in rows:
        out.setdefault(name, []).append(float(v))
    return out


def main():
    kernel, out_path, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = {"kernel": kernel, "counters": {}}
    for db in dbs:
        for name, vals in collect(db, kernel).items():
            if name.startswith("GRBM"):
                continue
            mx = max(vals) if vals else 0.0
            work = [v for v in vals if v > 0.5 * mx]
            res["counters"][name] = {"launches": len(vals), "mean_all": sum(vals) / max(len(vals), 1),
                                     "working_launches": len(work), "mean_working": sum(work) / max(len(work), 1)}
    cs = res["counters"]
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        for tag in ("all", "working"):
            f = cs["FETCH_SIZE"][f"mean_{tag}"] * 1024.0 * 2.0   # KiB -> B, gfx950 half-count correction
            w = cs["WRITE_SIZE"][f"mean_{tag}"] * 1024.0
            res[f"traffic_bytes_per_launch_{tag}"] = f + w
            res[f"fetch_bytes_per_launch_{tag}"] = f
            res[f"write_bytes_per_launch_{tag}"] = w
    # the kernel source the counters were collected on (bench.py flags a summary as stale when tl_gn.hip has changed since)
    import hashlib, os
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tloam_amd", "csrc", "tl_gn.hip")
    if os.path.exists(src):
        res["kernel_source_sha16"] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
