"""Rewrite profiles/<tag>_* from gpurun_out/ev_<tag>/ (tools/collect_evidence.sh): bench line, kernel stats, the step
breakdown, HBM traffic per kernel family from the FETCH_SIZE / WRITE_SIZE PMC passes and MFMA counters of the
convolution families.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-byte-per-lane
stores.  Per launch = summed counter / number of dispatches of the family in the PMC run (an eager bench.py step)."""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
EV = os.path.join(ROOT, "gpurun_out", f"ev_{tag}")
PR = os.path.join(ROOT, "profiles")

FAMILIES = {   # name -> regex on the demangled kernel name
    "conv_fwd": r"conv_igemm_kernel<\d+, \d+, \d+, \d+, 0, |conv_igemm_row3_kernel<\d+, \d+, \d+, 0[,>]|conv_stem_fwd_kernel|conv_igemm_stem_kernel",
    "dgrad": r"conv_igemm(_x4)?_kernel<\d+, \d+, \d+, \d+, 1, |conv_igemm_row3_kernel<\d+, \d+, \d+, 1[,>]",
    "dgrad+bn_reduce": r"conv_igemm(_x4)?_kernel<\d+, \d+, \d+, \d+, 3, |conv_igemm_row3_kernel<\d+, \d+, \d+, 3[,>]",
    "wgrad": r"conv_wgrad(_dma|_row3)?_kernel<|conv_stem_bwd_fused_kernel",
    "wgrad_reduce": r"wgrad_reduce(_v4)?_kernel",
    "bn_silu_apply": r"bn_silu_apply_kernel",
    "bn_silu_bwd_apply": r"bn_silu_bwd_apply_kernel",
}


def counter_rows(d):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def per_family(rows, counter):
    out = {}
    for fam, pat in FAMILIES.items():
        rx = re.compile(pat)
        tot, disp = 0.0, set()
        for r in rows:
            if r["Counter_Name"] == counter and rx.search(r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        out[fam] = (tot, len(disp))
    return out


def timeline():
    """idle gaps / queue hand-overs of one replayed step (tools/step_timeline.py) -> profiles/<tag>_step_timeline.txt"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), os.path.join(EV, "trace"), "6"],
                         capture_output=True, text=True).stdout
    open(os.path.join(PR, f"{tag}_step_timeline.txt"), "w").write(out)


def family_durations(bench):
    """rocprofv3 durations per family in two windows of the SAME trace: the last eager one-stream step (what bench.py's
    event timing measures: must agree with its avg_launch_us up to ~3 us of dispatch latency per launch) and one
    replayed step (kernels of two queues overlap and slow each other down)."""
    f = glob.glob(os.path.join(EV, "trace", "*kernel_trace.csv")) + glob.glob(os.path.join(EV, "trace", "*", "*kernel_trace.csv"))
    if not f:
        return
    tr = list(csv.DictReader(open(f[0]))); tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(tr) if "nchw_to_nhwc4" in r["Kernel_Name"]]
    if len(idx) < 8:
        return
    wins = {"eager": tr[idx[-2]:idx[-1]], "replay": tr[idx[len(idx) - 6]:idx[len(idx) - 5]]}
    fams = dict(FAMILIES); fams["wgrad"] = r"conv_wgrad(_dma|_row3)?_kernel<|conv_stem_bwd_fused_kernel|wgrad_reduce(_v4)?_kernel"
    bt = {r["family"]: r for r in bench.get("families", [])}
    lines = ["family                 launches(bench)  bench event-timed us/launch | rocprof eager step: us/launch (kernels) | rocprof replayed step: us/launch"]
    for fam, pat in fams.items():
        rx = re.compile(pat)
        if fam not in bt:
            continue
        n = bt[fam]["launches"]
        row = f"{fam:22s} {n:15d} {1e3 * bt[fam]['ms'] / n:28.2f}"
        for w in ("eager", "replay"):
            rs = [r for r in wins[w] if rx.search(r["Kernel_Name"])]
            tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
            row += f" | {tot / n:22.2f} ({len(rs)})"
        lines.append(row)
    open(os.path.join(PR, f"{tag}_family_durations.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def main():
    timeline()
    from object_detection_cib_amd import build as kb
    dig = kb.source_digest()
    bench = json.loads(open(os.path.join(EV, "bench.json")).read().strip().splitlines()[-1])
    fam_alg = {r["family"]: r for r in bench.get("families", [])}
    family_durations(bench)
    fr, wr = counter_rows(os.path.join(EV, "pmc_FETCH_SIZE")), counter_rows(os.path.join(EV, "pmc_WRITE_SIZE"))
    F, W = per_family(fr, "FETCH_SIZE"), per_family(wr, "WRITE_SIZE")
    fams, lines = {}, ["family                 launches   read MB/launch  write MB/launch  total MB/launch  algorithmic MB/launch  traffic/algorithmic"]
    for fam in FAMILIES:
        (f, nf), (w, nw) = F[fam], W[fam]
        if not nf or not nw:
            continue
        rd, wrb = 2 * f * 1024 / nf, w * 1024 / nw
        key = fam if fam in fam_alg else None
        alg = 1e6 * fam_alg[key]["algorithmic_MB"] / fam_alg[key]["launches"] if key else None
        fams[fam] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wrb, "traffic_bytes_per_launch": rd + wrb,
                     "dispatches": nf, "algorithmic_bytes_per_launch": alg}
        lines.append(f"{fam:22s} {nf:8d} {rd / 1e6:15.2f} {wrb / 1e6:16.2f} {(rd + wrb) / 1e6:16.2f} "
                     + (f"{alg / 1e6:22.2f} {(rd + wrb) / alg:20.2f}" if alg else f"{'-':>22s} {'-':>20s}"))
    json.dump({"csrc_digest": dig, "workload": bench.get("config", {}).get("workload", "yv5s B=64 640px") + "; one eager bench.py step per PMC pass",
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (--kernel-trace only); KiB units; "
                         "read = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B); mean over the family's dispatches",
               "families": fams}, open(os.path.join(PR, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(PR, f"{tag}_pmc_traffic_by_family.txt"), "w").write("\n".join(lines) + "\n")
    # MFMA counters
    mr = counter_rows(os.path.join(EV, "pmc_mfma"))
    if mr:
        tr = {}
        tf = glob.glob(os.path.join(EV, "pmc_mfma", "*kernel_trace.csv")) + glob.glob(os.path.join(EV, "pmc_mfma", "*/*kernel_trace.csv"))
        for r in csv.DictReader(open(tf[0])):
            tr[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        out = ["MFMA counters per kernel family (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
               "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE, one eager step; durations from the same pass, i.e. with counter overhead)",
               "FLOP = 512 x SQ_INSTS_VALU_MFMA_MOPS_BF16; TF/s = FLOP / sum of dispatch durations; MFMA pipe utilisation = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCD",
               "family                 launches   GFLOP/launch  us/launch    TF/s  of 2500   mfma busy cyc/launch  pipe util"]
        for fam, pat in FAMILIES.items():
            rx = re.compile(pat)
            acc, disp = {}, set()
            for r in mr:
                if rx.search(r["Kernel_Name"]):
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
            if not disp or not acc.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
                continue
            n = len(disp)
            ns = sum(tr.get(d, 0) for d in disp)
            flop = 512.0 * acc["SQ_INSTS_VALU_MFMA_MOPS_BF16"]
            cyc = acc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            util = acc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * cyc) if cyc else float("nan")
            tfs = flop / (ns * 1e-9) / 1e12 if ns else float("nan")
            out.append(f"{fam:22s} {n:8d} {flop / n / 1e9:13.2f} {ns / n / 1e3:10.1f} {tfs:7.0f} {tfs / 2500:8.3f} "
                       f"{acc.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n:22.0f} {util:10.3f}")
        open(os.path.join(PR, f"{tag}_mfma_counters.txt"), "w").write("\n".join(out) + "\n")
        print("\n".join(out))
    # bench line (traffic filled in for the family it names), kernel stats, step breakdown
    fam = bench["roofline"].get("family")
    if fam in fams:
        bench["roofline"]["traffic"] = round(fams[fam]["traffic_bytes_per_launch"])
    open(os.path.join(PR, f"bench_{tag}.json"), "w").write(json.dumps(bench) + "\n")
    ks = glob.glob(os.path.join(EV, "trace", "*kernel_stats.csv")) + glob.glob(os.path.join(EV, "trace", "*/*kernel_stats.csv"))
    shutil.copy(ks[0], os.path.join(PR, f"{tag}_kernel_stats.csv"))
    open(os.path.join(PR, f"{tag}_step_breakdown.txt"), "w").write(
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), os.path.join(EV, "trace")],
                       capture_output=True, text=True, cwd=ROOT).stdout)
    print("\n".join(lines))
    print(json.dumps(bench["roofline"]), bench["value"], bench["ms_per_step"])


if __name__ == "__main__":
    main()
