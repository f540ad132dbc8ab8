#!/usr/bin/env python3
"""Per-kernel summary (calls, average / total duration, share) of a rocprofv3 --kernel-trace run whose output is the
rocpd SQLite database (`*_results.db`) this ROCm version writes.  usage: rocpd_stats.py <dir or .db> [csv]"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
rows = {}
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    for d, s in zip(sorted(kd), sorted(ks)):
        for name, n, tot in c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {d} d join {s} s on d.kernel_id = s.id group by s.kernel_name"):
            a = rows.setdefault(name, [0, 0])
            a[0] += n
            a[1] += tot
def demangle(names):
    """kernel symbols -> readable names (llvm-cxxfilt from the ROCm tree; the '.kd' suffix of a kernel descriptor dropped)"""
    import shutil
    import subprocess

    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "c++filt"
    clean = [n[:-3] if n.endswith(".kd") else n for n in names]
    try:
        out = subprocess.run([tool], input="\n".join(clean), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out[: len(names)]))
    except Exception:
        return dict(zip(names, clean))


pretty = demangle(list(rows))
rows = {pretty[k]: v for k, v in rows.items()}
total = sum(v[1] for v in rows.values()) or 1
csv = len(sys.argv) > 2 and sys.argv[2] == "csv"
print("Name,Calls,TotalDurationNs,AverageNs,Percentage" if csv else f"{'kernel':90s} {'calls':>7s} {'avg us':>10s} {'share':>7s}")
for name, (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if csv:
        print(f'"{name}",{n},{tot},{tot / n:.1f},{100 * tot / total:.2f}')
    else:
        print(f"{name[:90]:90s} {n:7d} {tot / n / 1e3:10.1f} {100 * tot / total:6.1f}%")
