"""Last N kernel dispatches of a rocprofv3 rocpd results.db in launch order (name, grid in workgroups, duration, gap to the
previous kernel's end): one step of a launch-bound model.   python scripts/rocprof_timeline.py <results.db> [N=200] [match]"""
import sqlite3
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocprof_summary import short  # noqa: E402

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
match = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(path)
rows = list(c.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) from kernels order by start"))
rows = rows[-n:]
prev = None
for name, s, e, wgs in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = e
    if match and match not in name:
        continue
    print(f"{short(name, 70):<72}{wgs:>7}{(e - s) / 1e3:>9.1f} us  gap {gap:>7.1f}")
