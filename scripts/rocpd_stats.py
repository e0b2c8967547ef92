"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the per-kernel stats table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration (ns), percentage."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"']
    for n, calls, s, a, mn, mx in rows:
        lines.append(f'"{n}",{calls},{int(s)},{a:.1f},{int(mn)},{int(mx)},{100.0 * s / tot:.2f}')
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
