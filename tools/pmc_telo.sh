# development aid: instruction counts of tf_scan, the shift-and automaton (CORNETTO_TF_BP=0) against the bit planes (1), development build (rocprofv3 --pmc, no other tracing)
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CORNETTO_LIB=$R/cornetto_amd/libcornetto_hip_dev.so
for bp in 0 1; do
export CORNETTO_TF_BP=$bp
rm -rf $R/gpurun_out/pmc_telo
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/pmc_telo --output-format csv -- python3 $R/tools/perf_probe.py telo --mbases 1000 --reps 2 > /dev/null 2>&1
echo "== CORNETTO_TF_BP=$bp (1000 Mbases)"
python3 - <<PY
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob("$R/gpurun_out/pmc_telo/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.split(r"[<(]", r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::",""))[0][-40:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,c in agg.items():
    if "tf_scan" not in k: continue
    d=max(1,len(n[k]))
    print(k, "x%d"%d, {kk.replace("SQ_INSTS_",""): round(v/d/1e6,2) for kk,v in c.items()})
PY
done
rm -rf $R/gpurun_out/pmc_telo
