"""Summarise rocprofv3 --pmc passes into profiles/pmc_<workload>.json (per-kernel HBM traffic per launch).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py c4 gpurun_out/pmc_fetch gpurun_out/pmc_write [bench line of the counter run]

With the counter run's own bench line (its lengths differ from a default run's: fewer steps), the summary also records the
ALGORITHMIC bytes per launch of the dominant kernel in that run, so that traffic / algorithmic compares like with like.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (128-B requests
tallied at 64 B), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "mli::" in row["Kernel_Name"]:
                agg[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    workload, fetch_dir, write_dir = sys.argv[1:4]
    line = sys.argv[4] if len(sys.argv) > 4 else None
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    out = {"workload": workload, "unit": "bytes per launch",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; "
                     "FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(k, (0.0, 0))
        w_kib, nw = write.get(k, (0.0, 0))
        out["kernels"][k] = {"fetch_size_kib_raw": f_kib, "write_size_kib_raw": w_kib, "launches": max(nf, nw),
                             "read_bytes": 2 * f_kib * 1024, "write_bytes": w_kib * 1024,
                             "traffic_bytes": 2 * f_kib * 1024 + w_kib * 1024}
    if line and os.path.exists(line):
        rows = [ln for ln in open(line) if ln.startswith("{")]
        if rows:
            roof = json.loads(rows[-1]).get("roofline", {})
            out["counter_run"] = {"dominant_kernel": roof.get("kernel"),
                                  "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
                                  "note": "the dominant (lean scan) kernel's algorithmic bytes at the lengths of the counter run"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"pmc_{workload}.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:70]:70s} traffic {v['traffic_bytes'] / 1e6:10.1f} MB  (read {v['read_bytes'] / 1e6:.1f}, write {v['write_bytes'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
