#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel count / total / avg / min / max (us)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("""
        select s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z,
               count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name, 2, 3 order by 5 desc""").fetchall()
    total = sum(r[4] for r in rows)
    lines = ["kernel,grid_threads,block,calls,total_us,avg_us,min_us,max_us,pct"]
    for name, grid, blk, n, tot, mn, mx in rows:
        short = name.split("(")[0][:90]
        lines.append(f"{short},{grid},{blk},{n},{tot / 1e3:.1f},{tot / n / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * tot / total:.1f}")
    lines.append(f"TOTAL,,,{sum(r[3] for r in rows)},{total / 1e3:.1f},,,,100")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
