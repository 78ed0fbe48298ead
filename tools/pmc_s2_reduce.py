"""Reduce gpurun_out/pmc_s2/set*/p_counter_collection.csv (tools/pmc_s2.sh) into one table: per labelled launch group
(dispatch order: per layer REP x dgrad+bnred, REP x dgrad, REP x forward, REP x backward apply) the mean of every counter
and of the duration."""
import csv, glob, os, re, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "pmc_s2")
KINDS = ("dgrad+bnred", "dgrad", "fwd", "bwd_apply")
LAYERS = ("s1 32->64 @320", "s2 64->128 @160", "s3 128->256 @80")
INTEREST = re.compile(r"conv_igemm|conv_stem|bn_silu_bwd_apply")
table = collections.OrderedDict()
for sd in sorted(glob.glob(os.path.join(D, "set*"))):
    if not os.path.isdir(sd):
        continue
    cc = glob.glob(os.path.join(sd, "*counter_collection.csv"))
    kt = glob.glob(os.path.join(sd, "*kernel_trace.csv"))
    if not cc or not kt:
        continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    rows = list(csv.DictReader(open(cc[0])))
    per = collections.OrderedDict()
    for r in rows:
        d = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = [(k, v) for k, v in sorted(per.items()) if INTEREST.search(v["name"])]
    # run-length groups of identical kernel names, REP = 4 launches each
    groups, cur = [], []
    for k, v in seq:
        if cur and (cur[-1][1]["name"] != v["name"] or len(cur) == 4):
            groups.append(cur); cur = []
        cur.append((k, v))
    if cur:
        groups.append(cur)
    groups = [g for g in groups if len(g) == 4]
    for gi, g in enumerate(groups[:len(LAYERS) * len(KINDS)]):
        key = (LAYERS[gi // 4], KINDS[gi % 4])
        t = table.setdefault(key, {})
        t["us"] = sum(dur.get(str(k), 0) for k, _ in g[1:]) / 3e3
        for cname in g[0][1]:
            if cname != "name":
                t[cname] = sum(v.get(cname, 0.0) for _, v in g[1:]) / 3
cols = sorted({c for t in table.values() for c in t if c != "us"})
print("launch".ljust(34) + "us".rjust(9) + "".join(c.replace("SQ_", "").rjust(18) for c in cols))
for (layer, kind), t in table.items():
    print(f"{layer} {kind}".ljust(34) + f"{t['us']:9.1f}" + "".join(f"{t.get(c, float('nan')):18.4g}" for c in cols))
