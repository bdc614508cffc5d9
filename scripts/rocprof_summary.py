"""Condense a rocprofv3 rocpd results.db (kernel trace + stats) into a short text table.
usage: python scripts/rocprof_summary.py <results.db> [> profiles/rNN_xxx.txt]"""
import re
import sqlite3
import sys


def short(name, n=96):
    name = re.sub(r"\(.*", "", name)            # drop the argument list
    name = name.replace("void ", "").replace("cplxamd::", "")
    name = re.sub(r"at::native::\(anonymous namespace\)::|at::native::", "aten::", name)
    return name if len(name) <= n else name[: n - 3] + "..."


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), "
                          "max(end - start), max(vgpr_count), max(accum_vgpr_count), max(lds_size), "
                          "max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)) from kernels "
                          "group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats : {path.split('/')[-1]}   total kernel time {total/1e6:.3f} ms")
    print(f"{'kernel':<98}{'calls':>6}{'total_ms':>10}{'avg_us':>10}{'min_us':>9}{'max_us':>9}{'%':>7}"
          f"{'vgpr':>6}{'agpr':>6}{'lds':>8}{'wgs':>8}")
    for name, n, tot, avg, mn, mx, vg, ag, lds, wgs in rows:
        print(f"{short(name):<98}{n:>6}{tot/1e6:>10.3f}{avg/1e3:>10.1f}{mn/1e3:>9.1f}{mx/1e3:>9.1f}"
              f"{100*tot/total:>7.2f}{vg or 0:>6}{ag or 0:>6}{lds or 0:>8}{wgs or 0:>8}")


if __name__ == "__main__":
    main(sys.argv[1])
