"""Summarise one config directory of scripts/prof_round2.sh into a small JSON for profiles/.

Per DP kernel (name as rocprofv3 prints it): launches, average duration from --kernel-trace --stats, the SQ counters
(mean per launch), HBM bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, + WRITE_SIZE; both
in KiB), and derived: VALU wave-instructions per cell, LDS bank-conflict share, effective clock."""
import collections, csv, glob, json, os, sys

d, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
out = {"config": tag}
try:
    out["bench_line"] = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa
    out["bench_line"] = None
cells = (out["bench_line"] or {}).get("config", {}).get("cells_per_step_rank0")

stats = {}
for f in glob.glob(f"{d}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "opal_" in r["Name"]:
            stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                "pct": float(r["Percentage"])}
out["kernel_stats"] = stats

counters = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for sub in ("sq1", "sq2", "grbm", "fetch", "write"):
    for f in glob.glob(f"{d}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "opal_dp" not in k and "opal_rev" not in k and "opal_sw_stream" not in k and "opal_gl_stream" not in k and "opal_swap" not in k:
                continue
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), "lds": r.get("LDS_Block_Size"), "wg": r.get("Workgroup_Size"),
                       "grid": r.get("Grid_Size")}
pk = {}
for k, cs in counters.items():
    e = {n: sum(v) / len(v) for n, v in cs.items()}
    e["_launches_seen"] = max(len(v) for v in cs.values())
    e["_meta"] = meta.get(k)
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int((2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 4)
    if k in stats and e.get("GRBM_GUI_ACTIVE"):
        e["effective_clock_ghz_if_8xcd_sum"] = round(e["GRBM_GUI_ACTIVE"] / 8 / (stats[k]["avg_us"] * 1e3), 3)
        e["effective_clock_ghz_raw"] = round(e["GRBM_GUI_ACTIVE"] / (stats[k]["avg_us"] * 1e3), 3)
    pk[k] = e
out["per_kernel"] = pk
# searches of the profiled bench run = the calls of the DP kernel that runs once per search (the smallest call count among them:
# settle / warm-up / timed / clock-probe searches all launch it; the probe runs for >= 30 ms, so the count depends on the search)
calls = {k: stats.get(k, {}).get("calls", 0) for k in pk}
out["calls"] = calls
nz = [c for c in calls.values() if c > 0]
if cells and pk and nz:
    searches = int(sys.argv[4]) if len(sys.argv) > 4 else min(nz)
    out["searches_in_the_profiled_run"] = searches
    lps = {k: (calls[k] / searches if calls[k] else 0) for k in pk}
    out["launches_per_step"] = lps
    per_step = sum(e.get("SQ_INSTS_VALU", 0) * lps[k] for k, e in pk.items())
    out["valu_wave_instr_per_cell_lane"] = round(per_step * 64 / cells, 3)
    out["hbm_bytes_per_step"] = int(sum(e.get("hbm_bytes_per_launch", 0) * lps[k] for k, e in pk.items()))
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:600])
