"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass.  usage: pmc_sq_summary.py COUNTERS.csv OUT.json"""
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    if k.startswith("__amd"):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m["launches"] = len(next(iter(d.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        m["frac_wait_any"] = m.get("SQ_WAIT_ANY", 0.0) / wc
        m["frac_wait_inst"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
        m["frac_active"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    out[k] = m
json.dump(out, open(sys.argv[2], "w"), indent=1)
