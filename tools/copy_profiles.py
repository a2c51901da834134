"""copies what tools/collect_all.sh left under gpurun_out/ into profiles/ (run in the build container)"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
tr = {}
for w in ("config2", "config3", "config4"):
    s = json.load(open(f"gpurun_out/prof_{tag}_{w}_summary.json"))
    shutil.copy(f"gpurun_out/prof_{tag}_{w}_summary.json", f"profiles/{tag}_{w}_rocprof_summary.json")
    ks = sorted(glob.glob(f"gpurun_out/prof_{tag}_{w}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
    shutil.copy(ks, f"profiles/{tag}_{w}_kernel_stats.csv")
    dj = json.loads(open(f"gpurun_out/prof_{tag}_{w}.bench.json").read().strip().splitlines()[-1])
    tr[w] = {"batch": dj["roofline"].get("instances_per_launch", 4096), "hbm_bytes_per_launch": s["hbm_bytes_per_launch"], "fetch_size_raw_kib": s["FETCH_SIZE"], "write_size_kib": s["WRITE_SIZE"],
             "note": "FETCH_SIZE*1024*2 (gfx950 under-count of wide reads) + WRITE_SIZE*1024; separate --pmc passes; median over the launches of a 20-step run (traffic per launch does not depend on the step count); `batch` = instances per launch"}
    name = f"{tag}_bench_default.json" if w == "config4" else f"{tag}_bench_{w}.json"
    d = json.loads(open(f"gpurun_out/prof_{tag}_{w}.bench.json").read().strip().splitlines()[-1]); d["roofline"]["traffic"] = s["hbm_bytes_per_launch"]
    open("profiles/" + name, "w").write(json.dumps(d) + "\n")
    print(w, "%.3g steps/s" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "rocprof avg %.4f ms" % (float(s["kernel_stats"]["AverageNs"]) / 1e6),
          "traffic %.1f MB" % (s["hbm_bytes_per_launch"] / 1e6), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_all_cores", {}).get("value"))
json.dump(tr, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
shutil.copy("gpurun_out/phase_cycles.txt", f"profiles/{tag}_phase_cycles.txt")
shutil.copy("gpurun_out/parity_report.txt", f"profiles/{tag}_parity_report.txt")
for w in ("config5", "config4v"):
    line = open(f"gpurun_out/bench_{w}.json").read().strip().splitlines()[-1]
    open(f"profiles/{tag}_bench_{w}.json", "w").write(line + "\n"); d = json.loads(line)
    print(w, "%.3g steps/s" % d["value"], "ms %.3f" % d["ms_per_step"], "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_all_cores", {}).get("value"))
