"""When do the weight gradients run inside a replayed step?  usage: python tools/wg_schedule.py <rocprofv3 output dir>
Per step window of the kernel trace: queues used, when the loss ends, start of the first / second / median / last weight
gradient, end of the main chain (the stem's backward) and of the step - all in us from the step's first kernel."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
tr = list(csv.DictReader(open(f[0])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "nchw_to_nhwc4" in r["Kernel_Name"]]
for a, b in zip(idx[:-1], idx[1:]):
    st = tr[a:b]
    t0 = int(st[0]["Start_Timestamp"])
    us = lambda r, k: (int(r[k]) - t0) / 1e3
    wg = [us(r, "Start_Timestamp") for r in st if "conv_wgrad" in r["Kernel_Name"]]
    loss = [us(r, "End_Timestamp") for r in st if "loss_finalize" in r["Kernel_Name"]]
    stem = [us(r, "End_Timestamp") for r in st if "stem_bwd" in r["Kernel_Name"]]
    end = max(us(r, "End_Timestamp") for r in st)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st) / 1e3
    q = sorted(set(r["Queue_Id"] for r in st))
    if not wg:
        continue
    print("n=%d queues=%s loss_end=%.0f wgrad first=%.0f second=%.0f median=%.0f last=%.0f | main_end=%s step_end=%.0f sum=%.0f" % (
        len(st), ",".join(q), loss[0] if loss else -1, wg[0], wg[1], wg[len(wg) // 2], wg[-1],
        "%.0f" % stem[0] if stem else "-", end, busy))
