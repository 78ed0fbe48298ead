"""Formats the log of tools/first_epoch_sweep.sh into the HIP half of profiles/<tag>_first_epoch_samples.txt and appends the
CPU samples of the fixture (tests/golden/first_epoch.npz).
usage: tools/first_epoch_table.py <variant sweep log> <out file> [<ulp sweep log of tools/first_epoch_ulp.sh>]"""
import ast, re, sys
import numpy as np

log, out = sys.argv[1], sys.argv[2]
rows, cur = [], None
for line in open(log):
    m = re.match(r"== variant \[(.*)\] dx_fp32=(\d)", line)
    if m:
        cur = (m.group(1) or "default", "fp32" if m.group(2) == "1" else "bf16")
    elif line.startswith("fifths") and cur:
        fifths = ast.literal_eval(line[len("fifths "):line.index("]") + 1])
        rep = ast.literal_eval(line[line.index("{"):].strip())
        rows.append((cur[0], cur[1], fifths[-1], rep["map"], rep["map30"], rep["map50"]))
        cur = None
L = ["# First-epoch mAP samples, final kernels (tools/first_epoch_sweep.sh / first_epoch_ulp.sh on MI355X -> tools/first_epoch_table.py;",
     "# CPU rows: oracle/first_epoch.py --extra / --extra2 on the build host)",
     "# config: yv5s, 160 px, B=16, 500 steps, 8000 synthetic coco-zipf-like training images, 256 validation images", "#",
     "# HIP trainer: kernel summation-order variant | multi-producer dX accumulation | mean loss of the last fifth | map map30 map50"]
seen, stats = {}, {"bf16": [], "fp32": []}
for v, acc, last, m, m30, m50 in rows:
    key = (acc, last, m, m30, m50)
    note = ""
    if key in seen:
        note = f" (same trajectory as '{seen[key]}': the knob does not change this configuration)"
    else:
        seen[key] = v
        stats[acc].append((last, m, m50))
    L.append(f"{v:36s} | {acc} | {last:.4f} | {m:.4f} {m30:.4f} {m50:.4f}{note}")
for acc in ("bf16", "fp32"):
    a = np.array(stats[acc])
    L.append(f"# {acc} accumulation: {len(a)} distinct trajectories, last-fifth loss {a[:, 0].min():.4f} .. {a[:, 0].max():.4f}; "
             f"map mean {a[:, 1].mean():.4f} sd {a[:, 1].std(ddof=1):.4f}; map50 mean {a[:, 2].mean():.4f} sd {a[:, 2].std(ddof=1):.4f}")
ulp = []
if len(sys.argv) > 3:
    cur = None
    for line in open(sys.argv[3]):
        m = re.match(r"== variant \[(.*)\] dx_fp32=(\d)", line)
        if m:
            cur = m.group(1)
        elif line.startswith("fifths") and cur:
            fifths = ast.literal_eval(line[len("fifths "):line.index("]") + 1])
            rep = ast.literal_eval(line[line.index("{"):].strip())
            ulp.append((cur, fifths[-1], rep["map"], rep["map30"], rep["map50"]))
            cur = None
    L += ["#", "# HIP trainer, default kernels, one stem weight one bf16 ulp (2^-8) away from the seeded value (tools/first_epoch_ulp.sh):"]
    for v, last, m, m30, m50 in ulp:
        L.append(f"{v:36s} | bf16 | {last:.4f} | {m:.4f} {m30:.4f} {m50:.4f}")
    u = np.array([(r[2], r[4]) for r in ulp])
    L.append(f"# {len(u)} ulp draws: map mean {u[:, 0].mean():.4f} sd {u[:, 0].std(ddof=1):.4f}; map50 mean {u[:, 1].mean():.4f} sd {u[:, 1].std(ddof=1):.4f}")
g = np.load("tests/golden/first_epoch.npz", allow_pickle=True)
keys = [str(k) for k in g["map_keys"]]
S, tags = g["map_cpu_samples"], [str(t) for t in g["map_sample_tags"]]
L += ["#", "# CPU trainer (oracle/first_epoch.py; fixture tests/golden/first_epoch.npz): tag | map map30 map50"]
for t, r in zip(tags, S):
    L.append(f"{t:36s} | {r[keys.index('map')]:.4f} {r[keys.index('map30')]:.4f} {r[keys.index('map50')]:.4f}")
i50, im = keys.index("map50"), keys.index("map")
L.append(f"# {len(S)} CPU samples: map mean {S[:, im].mean():.4f} sd {S[:, im].std(ddof=1):.4f}; map50 mean {S[:, i50].mean():.4f} sd {S[:, i50].std(ddof=1):.4f}")
emu = np.array(["bf16" in t for t in tags])
welch = lambda a, b: (a.mean() - b.mean()) / np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
for name, sub in (("bf16-storage emulation", S[emu]), ("fp32", S[~emu])):
    L.append(f"#   {name}: {len(sub)} runs, map50 mean {sub[:, i50].mean():.4f} sd {sub[:, i50].std(ddof=1):.4f}")
h = np.array(stats["bf16"])[:, 2]
groups = [("the 8 kernel variants (bf16 accumulation)", h)]
if ulp:
    hu = np.array([r[4] for r in ulp])
    groups += [(f"the {len(hu)} ulp draws", hu), (f"all {len(h) + len(hu)} HIP trajectories", np.concatenate((h, hu)))]
for name, hh in groups:
    L.append(f"# HIP, {name}: map50 {hh.mean():.4f} sd {hh.std(ddof=1):.4f}; Welch t vs all {len(S)} CPU runs {welch(hh, S[:, i50]):+.2f}, "
             f"vs the {emu.sum()} emulation runs {welch(hh, S[emu][:, i50]):+.2f}, vs the {(~emu).sum()} fp32 runs {welch(hh, S[~emu][:, i50]):+.2f}")
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L[-6:]))
