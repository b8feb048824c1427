"""Turn what tools/collect_profiles.sh left under gpurun_out/ into the tracked summaries under profiles/<round>/."""
import collections, csv, glob, json, os, shutil, sys

dst = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01"
os.makedirs(dst, exist_ok=True)
for tag in "abfwtm":
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    out = {k: {c: v / n[(k, c)] for c, v in d.items()} for k, d in acc.items() if "lsr_" in k}
    json.dump(out, open(f"{dst}/pmc_pass_{tag}.json", "w"), indent=1, sort_keys=True)
stats = glob.glob("gpurun_out/prof_bench/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"{dst}/bench_kernel_stats.csv")
stats = glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"{dst}/train_iteration_kernel_stats.csv")
if os.path.exists("gpurun_out/prof_train.txt"):
    shutil.copy("gpurun_out/prof_train.txt", f"{dst}/train_iteration_host_phases.txt")
if os.path.exists("gpurun_out/bench_line.json"):
    shutil.copy("gpurun_out/bench_line.json", f"{dst}/bench_line.json")
f = json.load(open(f"{dst}/pmc_pass_f.json")); w = json.load(open(f"{dst}/pmc_pass_w.json"))
for k in f:
    if "emission" in k:
        t = json.load(open("profiles/emission_traffic.json"))
        t["kernel"] = k
        t["FETCH_SIZE_KB_per_launch"] = round(f[k]["FETCH_SIZE"]); t["WRITE_SIZE_KB_per_launch"] = round(w[k]["WRITE_SIZE"])
        t["hbm_bytes_per_launch"] = int((2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024)
        json.dump(t, open("profiles/emission_traffic.json", "w"), indent=1)
        print(k, t["hbm_bytes_per_launch"])
