"""copies what tools/collect_all.sh left under gpurun_out/ into profiles/ (run in the build container)"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
import csv
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rkfd_pkg
HEAD = rkfd_pkg.git_head()


def cal_factor():
    """known bytes / counter bytes of tools/ubench/traffic_cal.hip (8 B per lane rows of 30 doubles), read and write"""
    out = {}
    rows, nd = 1 << 21, 30
    known = 8.0 * rows * nd
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = glob.glob(f"gpurun_out/prof_{tag}_cal/{kind}/**/*counter_collection.csv", recursive=True)
        v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "rkfd_traffic_cal" in r["Kernel_Name"] and r["Counter_Name"] == ctr)
        out[kind] = known / (v[len(v) // 2] * 1024.0)
    return out


CAL = cal_factor()
print("calibration (known bytes / counter bytes): read %.3f write %.3f" % (CAL["fetch"], CAL["write"]))
tr = {}
for w in ("config2", "config3", "config4"):
    s = json.load(open(f"gpurun_out/prof_{tag}_{w}_summary.json"))
    ks = sorted(glob.glob(f"gpurun_out/prof_{tag}_{w}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
    shutil.copy(ks, f"profiles/{tag}_{w}_kernel_stats.csv")
    dj = json.loads(open(f"gpurun_out/prof_{tag}_{w}.bench.json").read().strip().splitlines()[-1])
    hb = s["FETCH_SIZE"] * 1024.0 * CAL["fetch"] + s["WRITE_SIZE"] * 1024.0 * CAL["write"]
    s["git_head_when_copied"] = HEAD
    if s.get("device_source_sha256") != rkfd_pkg.device_source_hash():
        print("WARNING: %s was collected on other device sources than this tree's (%s vs %s)" % (w, s.get("device_source_sha256"), rkfd_pkg.device_source_hash()))
    tr["device_source_sha256"] = s.get("device_source_sha256"); tr["git_head_when_copied"] = HEAD
    s["hbm_bytes_per_launch"] = hb; s["hbm_bytes_per_launch_note"] = "FETCH_SIZE*1024*%.4f + WRITE_SIZE*1024*%.4f (calibrated on tools/ubench/traffic_cal.hip; hbm_read_bytes_x2 is the uncalibrated 16 B-per-lane rule, kept for reference)" % (CAL["fetch"], CAL["write"])
    json.dump(s, open(f"profiles/{tag}_{w}_rocprof_summary.json", "w"), indent=1)
    tr[w] = {"batch": dj["roofline"].get("instances_per_launch", 4096), "hbm_bytes_per_launch": hb, "fetch_size_raw_kib": s["FETCH_SIZE"], "write_size_kib": s["WRITE_SIZE"],
             "calibration_read": CAL["fetch"], "calibration_write": CAL["write"],
             "note": "FETCH_SIZE*1024*cal_read + WRITE_SIZE*1024*cal_write; separate --pmc passes; median over the launches of the driver's command; the calibration "
                     "factors are known bytes / counter bytes of tools/ubench/traffic_cal.hip, which reads and writes instance-major rows of 30 doubles with 8 B per lane "
                     "like the step kernel (MI355X_MICROARCH.md: the x2 rule holds for 16 B-per-lane streams, other widths must be calibrated); `batch` = instances per launch"}
    name = f"{tag}_bench_default.json" if w == "config4" else f"{tag}_bench_{w}.json"
    d = json.loads(open(f"gpurun_out/prof_{tag}_{w}.bench.json").read().strip().splitlines()[-1]); d["roofline"]["traffic"] = s["hbm_bytes_per_launch"]
    open("profiles/" + name, "w").write(json.dumps(d) + "\n")
    print(w, "%.3g steps/s" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "rocprof avg %.4f ms" % (float(s["kernel_stats"]["AverageNs"]) / 1e6),
          "traffic %.1f MB" % (s["hbm_bytes_per_launch"] / 1e6), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_all_cores", {}).get("value"))
json.dump(tr, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
shutil.copy("gpurun_out/phase_cycles.txt", f"profiles/{tag}_phase_cycles.txt")
shutil.copy("gpurun_out/parity_report.txt", f"profiles/{tag}_parity_report.txt")
shutil.copy("gpurun_out/traffic_cal.txt", f"profiles/{tag}_traffic_calibration.txt")
open(f"profiles/{tag}_traffic_calibration.txt", "a").write("calibration (known bytes / counter bytes): read %.4f write %.4f\n" % (CAL["fetch"], CAL["write"]))
for f in ("ubench_pgs.txt",):
    if os.path.exists("gpurun_out/" + f):
        shutil.copy("gpurun_out/" + f, f"profiles/{tag}_" + f)
# the other bench lines of the collection, the Volume plugin's phase cycles and the kernel traces of config 5 / the Volume humanoid
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    name = os.path.basename(f)
    txt = open(f).read().strip().splitlines()
    if True:
        if txt:
            open(f"profiles/{tag}_{name}", "w").write(txt[-1] + "\n")
            d = json.loads(txt[-1]); print(name, "%.4g %s" % (d["value"], d["unit"]), d["config"].get("instances_per_wavefront"), d["roofline"].get("resident_instances_per_cu"))
if os.path.exists("gpurun_out/phase_cycles_volume.txt"):
    shutil.copy("gpurun_out/phase_cycles_volume.txt", f"profiles/{tag}_phase_cycles_volume.txt")
for w in ("config5", "config4_volume"):
    ks = sorted(glob.glob(f"gpurun_out/prof_{tag}_{w}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], f"profiles/{tag}_{w}_kernel_stats.csv")
        if os.path.exists(f"gpurun_out/prof_{tag}_{w}.bench.json"):
            shutil.copy(f"gpurun_out/prof_{tag}_{w}.bench.json", f"profiles/{tag}_{w}_kernel_stats.bench.json")
