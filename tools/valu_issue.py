#!/usr/bin/env python3
"""How much of a kernel's time is VALU issue?  From a rocprofv3 SQ counter summary (tools/pmc_sq_summary.py: SQ_INSTS_VALU, SQ_WAVES per launch) and the
kernel stats of the same workload (average duration): a wave64 instruction occupies its SIMD's 16-lane VALU for 4 cycles (MI355X_MICROARCH.md), a CU has 4
SIMDs, the chip 256 CUs at 2.4 GHz, so   t_valu = INSTS_VALU * 4 cycles / (1024 SIMDs * 2.4 GHz)   is the time the kernel would take if it did nothing but
issue its vector instructions, perfectly spread.  t_valu / t_kernel close to 1 = bound by instruction issue, not by memory.
usage: valu_issue.py PMC_SQ_SUMMARY.json KERNEL_STATS.csv [OUT.json] [PROBE.log]
PROBE.log (tools/chan_probe.py --kernels): the kernels' ISOLATED times (HIP events, each kernel alone on the chip) replace the in-step averages of the
stats file, in which concurrent kernels of the step's four streams share the wave slots."""
import csv, json, sys
sq = json.load(open(sys.argv[1]))
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Name"].split("(")[0].replace("void ", "").strip()] = float(r["AverageNs"]) * 1e-3
if len(sys.argv) > 4:
    t = open(sys.argv[4]).read()
    iso = json.loads(t[t.find("{"):])["times"]
    alias = {"k_flux_hor": "k_flux_hor_fused", "k_flux_hor_nt": "k_flux_hor_fused", "k_diff_flux_nt": "k_diff_flux", "k_kpp_smooth": "k_kpp_smooth1", "k_kpp_smooth_u": "k_kpp_smooth1",
             "k_vert_vel": "k_vert_vel_hbar", "k_pgf_tile": "k_pgf", "k_edge_transport_tile": "k_edge_transport", "k_bolus": "bolus_add", "k_tr_grad_elem_b": "k_tr_grad_elem"}
    for k in list(sq):
        b = k.split("<")[0]
        b = alias.get(b, b)
        if b in iso:
            dur[k] = iso[b] * 1e6
rows = []
for k, v in sq.items():
    if k not in dur or not v.get("SQ_INSTS_VALU"):
        continue
    t_valu = v["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e6
    rows.append({"kernel": k, "us": round(dur[k], 1), "valu_insts_per_wave": round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1.0), 0), "waves": int(v["SQ_WAVES"]),
                 "valu_issue_us": round(t_valu, 1), "valu_issue_frac": round(t_valu / dur[k], 3), "frac_wait_any": round(v.get("frac_wait_any", 0.0), 3)})
rows.sort(key=lambda r: -r["us"])
print(f"{'kernel':34s} {'us':>8s} {'insts/wave':>10s} {'VALU us':>8s} {'VALU frac':>9s}")
for r in rows:
    if r["us"] >= 20.0:
        print(f"{r['kernel'][:34]:34s} {r['us']:8.1f} {r['valu_insts_per_wave']:10.0f} {r['valu_issue_us']:8.1f} {r['valu_issue_frac']:9.2f}")
if len(sys.argv) > 3:
    json.dump({"model": "t_valu = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * 2.4 GHz)", "kernels": rows}, open(sys.argv[3], "w"), indent=1)
