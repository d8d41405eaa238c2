"""Print the start/end times (us) of every kernel of one model step from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys

def main():
    pat = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    files = glob.glob(pat, recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
    rows.sort()
    # a step starts at k_vel_nodes
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k_vel_nodes")]
    if len(idx) <= step + 1:
        step = len(idx) // 2
    a, b = idx[step], idx[step + 1]
    t0 = rows[a][0]
    lo = max(0, a - 3)
    for s, e, n, q in rows[lo:b + 1]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q:>3}  {n}")
    print("step length us:", (rows[b][0] - t0) / 1e3)

main()
