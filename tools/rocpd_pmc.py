#!/usr/bin/env python3
"""Per-kernel averages of a PMC counter from a rocprofv3 rocpd SQLite database."""
import sqlite3, sys, json


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "--schema" in sys.argv:
        for t in tabs:
            if "pmc" in t.lower() and "_0" not in t:
                print(t, [r[1] for r in cur.execute(f"pragma table_info({t})")])
        return
    rows = cur.execute("""
        select s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, p.name, count(*), avg(e.value), min(e.value), max(e.value)
        from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
             join rocpd_kernel_dispatch d on e.event_id = d.event_id
             join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by 1, 2, 3 order by 4 desc""").fetchall()
    res = []
    for name, grid, ctr, n, avg, mn, mx in rows:
        res.append({"kernel": name.split("(")[0][:80], "grid_threads": grid, "counter": ctr, "launches": n, "avg": avg, "min": mn, "max": mx})
        print(f"{name.split('(')[0][:70]:70s} grid {grid:8d} {ctr:12s} n={n:6d} avg={avg:14.1f} min={mn:12.1f} max={mx:12.1f}")
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None)
