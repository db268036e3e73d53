"""Aggregate rocprofv3 --pmc counter_collection.csv files (any number of passes) into one CSV per kernel family:
kernel, grid_threads, counter, mean_per_dispatch, dispatches -- profiles/<round>/pmc_<family>.csv.

    python scripts/pmc_kernels.py OUT_DIR PASS_DIR [PASS_DIR ...]
Families: conv3x3_sp_kernel (pipelined LDS-DMA), conv3x3_patch_kernel (register-staged A/B reference), conv_wgrad (all-taps
body `conv_wgrad_kernel<9, true>` and the tap-split A/B reference `conv_wgrad_tapsplit_kernel`), conv_fprop_kernel."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16

FAMILIES = {"conv3x3_sp_kernel": "conv3x3_sp_kernel", "conv3x3_patch_kernel": "conv3x3_patch_kernel",
            "conv3x3_dw_kernel": "conv3x3_dw_kernel", "conv_wgrad": "conv_wgrad",
            "conv_fprop_kernel": "conv_fprop_kernel", "quantize_mx8_kernel": "quantize_mx8_kernel"}


def main(out_dir, *pass_dirs):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for d in pass_dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                name = re.sub(r"\(.*$", "", name).strip()
                grid = row.get("Grid_Size") or row.get("Grid_Size_X") or ""
                a = acc[(name, grid, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    os.makedirs(out_dir, exist_ok=True)
    stamp = f"# kernel sources sha16 {kernel_source_sha16()} commit {os.environ.get('RGBD_COMMIT', '?')}"
    for fam, key in FAMILIES.items():
        rows = sorted((k, v) for k, v in acc.items() if key in k[0])
        if not rows:
            continue
        with open(os.path.join(out_dir, f"pmc_{fam}.csv"), "w") as f:
            f.write(stamp + "\n")
            w = csv.writer(f)
            w.writerow(["kernel", "grid_threads", "counter", "mean_per_dispatch", "dispatches"])
            for (name, grid, counter), (tot, n) in rows:
                w.writerow([name, grid, counter, round(tot / n, 1), n])
        print(fam, len(rows), "rows")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
