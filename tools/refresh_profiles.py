"""Rewrite profiles/ from the latest gpurun_out/ collection (bench line, rocprof stats, PMC passes, step breakdown)."""
import csv, json, re, subprocess, sys
sys.path.insert(0, '.')
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = subprocess.run([sys.executable, "tools/pmc_traffic.py", "gpurun_out/pmc_fetch", "gpurun_out/pmc_write", "conv_igemm_kernel", ", 0, "],
                     capture_output=True, text=True).stdout
open(f"profiles/{tag}_pmc_traffic_conv_fwd.txt", "w").write(out)
rd = float(re.search(r"read\s+([\d.]+) MB", out).group(1)) * 1e6
wr = float(re.search(r"write\s+([\d.]+) MB", out).group(1)) * 1e6
nd = int(re.search(r": (\d+) / \d+ dispatches", out).group(1))
json.dump({"kernel": "conv_igemm_kernel<MODE_RAW> (forward conv, all 57 layers)", "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
           "traffic_bytes_per_launch": rd + wr,
           "method": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), KiB units, FETCH_SIZE x2 (gfx950), mean over {nd} dispatches of bench.py --no-graph --steps 2 --warmup 1",
           "workload": "yv5s B=64 640px"}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
rows = list(csv.DictReader(open("gpurun_out/prof_r01_final/r01_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fw = [r for r in rows if re.search(r"conv_igemm_kernel<\d+, \d+, \d+, \d+, 0, ", r["Name"])]
calls = sum(int(r["Calls"]) for r in fw); t = sum(float(r["TotalDurationNs"]) for r in fw)
b = json.loads(open("gpurun_out/bench_r01_final.json").read().strip().splitlines()[-1])
b["roofline"]["traffic"] = round(rd + wr)
open(f"profiles/bench_{tag}_final.json", "w").write(json.dumps(b) + "\n")
txt = ("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (summary csv: profiles/%s_kernel_stats.csv)\n"
       "forward conv family conv_igemm_kernel<..., MODE_RAW, ...>: %d calls, average %.2f us per launch, %.1f%% of GPU kernel time\n"
       "bench.py roofline.avg_launch_us (HIP events around each launch of one eager step): %.2f us  (profiles/bench_%s_final.json)\n"
       % (tag, calls, t / calls / 1e3, 100 * t / tot, b["roofline"]["avg_launch_us"], tag))
open(f"profiles/{tag}_fwd_conv_family.txt", "w").write(txt)
subprocess.run(["cp", "gpurun_out/prof_r01_final/r01_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv"])
open(f"profiles/{tag}_step_breakdown.txt", "w").write(
    subprocess.run([sys.executable, "tools/prof_summary.py", "gpurun_out/prof_r01k"], capture_output=True, text=True).stdout)
print(txt, out, json.dumps(b["roofline"]), b["value"], b["ms_per_step"], b.get("cpu_baseline"))
