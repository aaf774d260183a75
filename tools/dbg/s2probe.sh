cd /tmp; export TMPDIR=/tmp
for v in "" "SUP3R_AMD_NO_MASK_FUSE=1"; do
  rm -rf /tmp/s2p
  env $v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/s2p -- python /root/repo/tools/train_probe.py --gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4 --precision bf16 --iters 2 > /dev/null 2>&1
  python - "$v" <<PY
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/s2p/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "dgrad_s2_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(sys.argv[1] or "default", k, "FETCH x2 GB %.3f"%(2*sum(v)/len(v)/1e6), len(v))
PY
done
