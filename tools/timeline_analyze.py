"""Timeline of the step from a rocprofv3 --kernel-trace CSV: per step (delimited by the first kernel of the step) the wall span,
the sum of kernel durations, the time at least one kernel runs (union) and the idle time; then the mean start offset / duration per
kernel name within a step.  Usage: python tools/timeline_analyze.py <kernel_trace.csv> <first_kernel_substring> [out.json]"""
import csv, sys, json, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
key = sys.argv[2]
starts = [i for i, r in enumerate(rows) if key in r[2]]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1) if starts[i + 1] - starts[i] >= 20]     # (not the per-kernel timing loops)
steps = steps[len(steps) // 2:]            # the second half: warmed up
span = []; ssum = []; union = []
per = collections.defaultdict(list)
for a, b in steps:
    ks = rows[a:b]
    t0 = ks[0][0]; t1 = max(k[1] for k in ks)
    span.append(rows[b][0] - t0); ssum.append(sum(k[1] - k[0] for k in ks))
    ev = sorted([(k[0], 1) for k in ks] + [(k[1], -1) for k in ks]); cur = 0; last = None; u = 0
    for t, d in ev:
        if cur > 0: u += t - last
        cur += d; last = t
    union.append(u)
    seen = collections.Counter()
    for k in ks:
        seen[k[2]] += 1
        per[(k[2].split("(")[0][:40], seen[k[2]], k[3])].append((k[0] - t0, k[1] - k[0]))
n = len(steps)
out = {"steps": n, "span_us": sum(span) / n / 1e3, "sum_kernel_us": sum(ssum) / n / 1e3, "union_us": sum(union) / n / 1e3,
       "idle_us": (sum(span) - sum(union)) / n / 1e3, "kernels_per_step": sum(b - a for a, b in steps) / n}
print(json.dumps(out))
tab = sorted(((sum(x[0] for x in v) / len(v) / 1e3, sum(x[1] for x in v) / len(v) / 1e3, k) for k, v in per.items() if len(v) >= n // 2))
prev_end = 0.0
for s, d, k in tab:
    print(f"{s:8.1f} {d:7.1f}  end {s + d:8.1f}  q{k[2]:>3s}  {k[0]}#{k[1]}")
if len(sys.argv) > 3:
    json.dump({"summary": out, "kernels": [{"start_us": s, "dur_us": d, "name": k[0], "occurrence": k[1], "queue": k[2]} for s, d, k in tab]}, open(sys.argv[3], "w"), indent=0)
