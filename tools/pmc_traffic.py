#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) of
`bench.py` into profiles/<tag>_traffic.json: per kernel, launches and HBM bytes per launch.

Units and corrections as MI355X_MICROARCH.md section HBM prescribes: both counters are in KiB;
on gfx950 FETCH_SIZE reads exactly half of the bytes of a wide coalesced streaming read, so the
read side is doubled (our kernels read with 4..16-byte-per-lane coalesced loads; k_byte_hist, a
pure 16-B/lane stream of known size, calibrates the factor: see 'calibration' in the output);
WRITE_SIZE is exact for streaming stores.

    python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json> [workload] [n] [builds per pass]
"""
import collections
import csv
import glob
import json
import sys


def load(counter_dir):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(counter_dir + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            full = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sa::", "")
            k = full.split("<")[0]
            if k in ("k_scatter_windows",):
                k = "k_scatter_pairs"                 # bench.py's kernel class of all binned ISA writes
            if k == "k_onesweep":                    # template arguments: <threads, items, KEY TYPE, values through the keys' buffer, workgroups per CU>
                k = "k_onesweep32" if "unsigned int" in full else "k_onesweep"
            if k in ("k_group_sort_straddle", "k_group_sort_big"):
                k = "k_group_sort"                    # bench.py's class of all fused gather + group-sort kernels
            if k.startswith("k_radix_downsweep"):     # all tile-scatter variants are one kernel class per key width in bench.py
                # template arguments: <threads, items, granule, min waves, stamps, KEY TYPE, prefetched items>
                k = "k_radix_downsweep32" if "unsigned int" in full else "k_radix_downsweep"
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    root, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else None
    n = int(sys.argv[4]) if len(sys.argv) > 4 else None
    builds = int(sys.argv[5]) if len(sys.argv) > 5 else None      # builds each PMC pass ran (bench.py --steps 2 --warmup 1: 1 + 2 + 1 profiled = 4)
    fetch, write = load(root + "/pmc_FETCH_SIZE"), load(root + "/pmc_WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fl, fv = fetch.get(k, [0, 0.0])
        wl, wv = write.get(k, [0, 0.0])
        launches = max(fl, wl)
        if not launches or not k.startswith("k_"):
            continue
        rd = 2.0 * fv * 1024 / launches
        wr = wv * 1024 / launches
        kernels[k] = {"launches": launches, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                      "hbm_bytes_per_launch": round(rd + wr)}
        if builds:
            kernels[k]["hbm_bytes_per_build"] = round((rd + wr) * launches / builds)
    cal = None
    if n and "k_byte_hist" in kernels:
        cal = {"kernel": "k_byte_hist", "known_read_bytes": n,
               "FETCH_SIZE_x2_bytes": kernels["k_byte_hist"]["read_bytes_per_launch"]}
    json.dump({"workload": workload, "n_bytes": n, "builds_per_pass": builds, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
               "passes; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 FETCH_SIZE halving corrected)",
               "calibration": cal, "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
