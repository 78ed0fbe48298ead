"""Per-layer conv timing table from a rocprofv3 kernel trace of bench.py --no-graph (last full step)."""
import csv, glob, sys
sys.path.insert(0, '.')
from object_detection_cib_amd.engine.graph import build_graph
d = sys.argv[1]
tr = list(csv.DictReader(open((glob.glob(d + '/*/*kernel_trace.csv') + glob.glob(d + '/*kernel_trace.csv'))[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'nchw_to_nhwc4' in r['Kernel_Name']]
step = tr[idx[-2]:idx[-1]]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
g = build_graph(3, 10, 0.5, 0.33)
units = [op.unit for op in g.ops if op.kind == 'conv']
B = 64
isk = lambda r, s: s in r['Kernel_Name']
import re
def mode(r):
    m = re.search(r'conv_igemm(?:_x4)?_kernel<\d+, \d+, \d+, \d+, (\d), ', r['Kernel_Name'])
    if m:
        return int(m.group(1))
    m = re.search(r'conv_igemm_row3_kernel<\d+, \d+, \d+, (\d)>', r['Kernel_Name'])
    return int(m.group(1)) if m else -1
fw = [r for r in step if mode(r) == 0]
dg = [r for r in step if mode(r) in (1, 3)]
wg = [r for r in step if isk(r, 'conv_wgrad')]
# backward per unit in reverse order; the three heads come first (one dgrad + one wgrad each)
ru = list(reversed(units))
dmap, wmap = {}, {}
dpos, wpos = 3, 3
dual_shorts = {u.sibling.name for u in units if u.sibling is not None}     # their dX comes from the main_conv's launch
for u in ru:
    if not u.stem and u.name not in dual_shorts:
        dmap[u.name] = dur(dg[dpos]); dpos += 1
    wmap[u.name] = dur(wg[wpos]); wpos += 1
tot = [0, 0, 0]
for i, u in enumerate(units):
    st = u.dst.stride; Ho = 640 // st; M = B * Ho * Ho
    cin = 3 if u.stem else u.cin; k = 6 if u.stem else u.k
    Min = B * (640 // (1 if u.stem else u.src.stride)) ** 2
    byts = 2 * (Min * cin + M * u.cout); fl = 2 * M * u.cout * cin * k * k
    f = dur(fw[i]); tot[0] += f
    line = "%-44s %4d %4d %d%d %7d | %6.1f %5.0f %4.0f" % (u.name[-44:], cin, u.cout, k, u.s, M, f, byts / f / 1e3, fl / f / 1e6)
    if u.name in dmap:
        dd = dmap[u.name]; tot[1] += dd; line += " | %6.1f %5.0f %4.0f" % (dd, byts / dd / 1e3, fl / dd / 1e6)
    else:
        line += " |    -     -    -  "
    w = wmap[u.name]; tot[2] += w; line += " | %6.1f %5.0f %4.0f" % (w, byts / w / 1e3, fl / w / 1e6)
    print(line)
print("totals fwd/dgrad/wgrad us", [round(t) for t in tot])
fam = {}
for r in step:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    key = n.split('(')[0][:48]
    fam.setdefault(key, [0, 0.0]); fam[key][0] += 1; fam[key][1] += dur(r)
print("step kernel time %.2f ms" % (sum(v[1] for v in fam.values()) / 1e3))
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:60]:
    print("   %8.1f us  x%-4d %s" % (t, c, k))
