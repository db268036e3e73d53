"""Which kernels of the OTHER queue are in flight while a named kernel runs?  (from a rocprofv3 --kernel-trace CSV)
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 6 --warmup 10 --no-cpu-baseline --no-roofline
  python scripts/trace_overlap.py DIR/**/t_kernel_trace.csv planes_outer_kernel<4> from_planes_kernel<4> linear_fwd adain_reduce
For every instance of a named kernel in the last complete step: its duration, and the kernels of other queues whose
[start, end] intersects its own, with the overlapped share of the instance's duration.  A kernel that a trace shows at 8x
its stand-alone time with a chip-filling kernel of the other queue covering 100 % of it is stretched, not slow."""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
names = sys.argv[2:]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "zero_multi" in r["Kernel_Name"]]
step = rows[starts[-3]:starts[-2]]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:60]


queues = sorted({r["Queue_Id"] for r in step})
print(f"step of {len(step)} kernels on queues {queues}")
summary = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in step:
    k = short(r["Kernel_Name"])
    if not any(n in k for n in names):
        continue
    dur = r["e"] - r["s"]
    others = []
    covered = 0
    for o in step:
        if o["Queue_Id"] == r["Queue_Id"] or o["e"] <= r["s"] or o["s"] >= r["e"]:
            continue
        ov = min(o["e"], r["e"]) - max(o["s"], r["s"])
        covered += ov
        others.append(f"{short(o['Kernel_Name'])} ({100.0 * ov / dur:.0f} %, itself {(o['e'] - o['s']) / 1e3:.0f} us)")
    s = summary[k]
    s[0] += 1
    s[1] += dur / 1e3
    s[2] += min(covered, dur) / 1e3
    print(f"{k:58s} q{r['Queue_Id']} {dur / 1e3:7.1f} us | other queue: " + ("; ".join(others) if others else "idle"))
print()
for k, (n, d, c) in summary.items():
    print(f"{k:58s} {n:3d} instances, {d / n:7.1f} us average, {100.0 * c / d:5.1f} % of it under a kernel of another queue")
