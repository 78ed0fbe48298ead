R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_deep"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export BENCH_CONV_ONLY=fd
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/$c" -o p -- python3 "$R/tools/bench_conv.py" "s4.main 512->256 1x1 @20" "s4.conv1 256->256 1x1 @20" "s4.b.conv2 256->256 3x3 @20" "s3.conv1 128->128 1x1 @40" "s3.b.conv2 128->128 3x3 @40" > "$O/$c.log" 2>&1 || tail -3 "$O/$c.log"
done
find "$O" -name "*.db" -delete
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/pmc_deep"
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(O+f"/{c}/**/*counter_collection.csv",recursive=True)
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"]!=c: continue
        k=(r["Kernel_Name"][:70], r["Grid_Size"])
        agg.setdefault(k,[]).append(float(r["Counter_Value"]))
    print("==",c,"(KiB per dispatch; FETCH_SIZE x2 = bytes read)")
    for k,v in agg.items():
        if "conv_igemm" in k[0]: print(f"{k[0]:72s} grid {k[1]:>8s} n={len(v):3d} avg {sum(v)/len(v):10.1f} KiB")
PY
