#!/usr/bin/env python3
"""Reads the two rocprofv3 --pmc result databases (one pass FETCH_SIZE, one pass WRITE_SIZE; the TCC
block cannot hold both) and writes profiles/pmc_traffic.json: HBM-side bytes per launch of the
library's kernels.  FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes, so it is doubled (exact for 16-byte-per-lane streaming reads; this library's
reads are 12-byte-per-lane loads and 16-byte LDS-DMA, so read it as an estimate).

    pmc_traffic.py fetch_results.db write_results.db out.json
"""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = ? "
                       "and name like '%pqd::%' group by name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pqd::", "")
    return n.replace("<", "_").replace(">", "").replace(", ", "_")


def main(fetch_db, write_db, out):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1",
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-byte read requests as 64 bytes); WRITE_SIZE KiB x 1024",
           "kernels": {}}
    for name in sorted(set(f) | set(w)):
        fb = f.get(name, (0, 0.0))[1] * 1024 * 2
        wb = w.get(name, (0, 0.0))[1] * 1024
        k = short(name)
        res["kernels"][k] = {"launches": f.get(name, (0, 0))[0], "fetch_bytes_per_launch": round(fb),
                             "write_bytes_per_launch": round(wb), "bytes_per_launch": round(fb + wb)}
        res[k + "_bytes_per_launch"] = round(fb + wb)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:32s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB  x{v['launches']}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
