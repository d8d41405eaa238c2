"""Aggregate two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only) into the
per-kernel HBM-side traffic table bench.py reports as roofline.traffic.  Units and corrections as MI355X_MICROARCH.md
(HBM / rocprofv3): both counters are in KiB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B -> doubled (checked
here on k_tr_ab, whose traffic is known exactly: it reads tr_arr + tr_arr_old and writes tr_arr_old); WRITE_SIZE is exact.
usage: pmc_summary.py FETCH.csv WRITE.csv OUT.json [WORKLOAD_KEY]   (the key bench.py matches its workload against)"""
import collections, csv, json, sys


def per_kernel(fn, tag):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == tag:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


def main():
    f, w, out = sys.argv[1:4]
    wkey = sys.argv[4] if len(sys.argv) > 4 else "pi_pp"
    F, W = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    tab = {}
    for k in sorted(set(F) | set(W)):
        if k.startswith("__amd"):
            continue
        fv, wv = F.get(k, [0.0]), W.get(k, [0.0])
        # per-tracer kernels are launched for one tracer (bench.py's per-kernel timing) and for both (the step): min / max
        tab[k] = {"launches": len(fv), "fetch_KiB_min": min(fv), "fetch_KiB_max": max(fv), "write_KiB_min": min(wv), "write_KiB_max": max(wv),
                  "traffic_bytes_min": (2.0 * min(fv) + min(wv)) * 1024.0, "traffic_bytes_max": (2.0 * max(fv) + max(wv)) * 1024.0}
    json.dump({"workload": wkey, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) around bench.py --steps 60 --warmup 20; templated kernels keep their arguments (<false> = without Redi, <true> = with: the other_physics leg)",
               "correction": "traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes); gfx950 FETCH_SIZE counts 128-B requests as 64 B",
               "kernels": tab}, open(out, "w"), indent=1)
    ab = tab.get("k_tr_ab")
    if ab:
        print("calibration k_tr_ab (both tracers): corrected fetch %.0f KiB, write %.0f KiB" % (2 * ab["fetch_KiB_max"], ab["write_KiB_max"]))


main()
