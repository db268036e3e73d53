"""Per-queue timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: what runs where, when, and the idle gaps.
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 6 --warmup 10 --no-cpu-baseline --no-roofline
  python scripts/timeline.py DIR/**/t_kernel_trace.csv [--full]
A step starts at its zero_multi_kernel launch (first kernel of the prep phase)."""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "zero_multi" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = step[0]["s"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]


print(f"step: {len(step)} kernels, {(max(r['e'] for r in step) - t0) / 1e6:.3f} ms from first launch to last end")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, first start {(rs[0]['s'] - t0) / 1e6:.3f} ms, last end "
          f"{(max(r['e'] for r in rs) - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
    prev = None
    for r in rs:
        gap = (r["s"] - prev) / 1e3 if prev else 0.0
        if "--full" in sys.argv or gap > 20 or (r["e"] - r["s"]) > 150e3:
            print(f"   {(r['s'] - t0) / 1e6:7.3f} ms  +{(r['e'] - r['s']) / 1e3:7.1f} us  idle before {gap:7.1f} us  {short(r['Kernel_Name'])}")
        prev = max(prev or 0, r["e"])
