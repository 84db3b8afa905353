#!/usr/bin/env python3
"""Reads the counter CSVs of two rocprofv3 --pmc passes (one FETCH_SIZE, one WRITE_SIZE: the TCC block
cannot hold both) and writes profiles/pmc_traffic.json: HBM-side bytes per launch of the library's
kernels, with the digest of the kernel sources they were measured on (bench.py reports the traffic
only while the sources still have that digest).  FETCH_SIZE and WRITE_SIZE are reported in KiB; on
gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so it is doubled (exact for 16-byte-per-lane
streaming reads; this library's reads are 12-byte-per-lane loads and 16-byte LDS-DMA, so read it as an
estimate).

    pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_digest():
    """sha256 over the kernel sources (the same function as in bench.py)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "parquet-java_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "pqd::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pqd::", "")
    return n.replace("<", "_").replace(">", "").replace(", ", "_")


def main(fetch_csv, write_csv, out):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1 --no-verify",
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-byte read requests as 64 bytes); WRITE_SIZE KiB x 1024",
           "sources_sha256": sources_digest(), "kernels": {}}
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
