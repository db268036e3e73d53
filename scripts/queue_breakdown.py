"""Launches and kernel time per HW queue from a rocprofv3 --kernel-trace CSV (which stream carries what)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
by = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", r["Kernel_Name"])[:70]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = by[r[qkey]][n]
    e[0] += 1; e[1] += d
for q, ks in by.items():
    tot_n = sum(v[0] for v in ks.values()); tot_t = sum(v[1] for v in ks.values())
    print(f"== queue {q}: {tot_n / steps:.0f} launches/step, {tot_t / steps / 1e6:.2f} ms/step")
    for n, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"   {c / steps:6.1f} x {t / c / 1e3:7.1f} us  {n}")
