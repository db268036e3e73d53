"""Aggregate rocprofv3 --pmc counter_collection.csv files (FETCH_SIZE and WRITE_SIZE passes) into
profiles/<round>/bench_pmc_traffic_<workload>.json: HBM bytes per launch for every kernel of ONE bench.py workload
(bench.py:workload_key; bench.py only quotes a profile whose recorded workload is the one it is running).

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts 128-byte requests as 64 B, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KB)."""
import collections
import csv
import glob
import json
import re
import sys


def load(pattern, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*$", "", name).strip()
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main(fetch_dir, write_dir, out, command, workload=None):
    f = load(fetch_dir + "/**/*counter_collection.csv", "FETCH_SIZE")
    w = load(write_dir + "/**/*counter_collection.csv", "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        n = max(f[name][1], w[name][1], 1)
        fk, wk = f[name][0] / max(f[name][1], 1), w[name][0] / max(w[name][1], 1)
        kernels[name] = {"fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "launches": n}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha16
    json.dump({"command": command, "workload": workload, "source_sha16": kernel_source_sha16(), "commit": os.environ.get("RGBD_COMMIT"),
               "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B requests "
                             "as 64 B; unit KB)",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k in ("conv3x3_sp_kernel<128, false, 0, 0>", "conv_wgrad_multi_kernel<9, true>"):
        if k in kernels:
            print(k, kernels[k])


if __name__ == "__main__":
    main(*sys.argv[1:6])
