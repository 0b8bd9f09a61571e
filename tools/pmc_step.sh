# development aid: instruction counts of every kernel of the step, one stage after the other (tools/perf_probe.py all) (rocprofv3 --pmc, no other tracing)
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_step
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/pmc_step --output-format csv -- python3 $R/tools/perf_probe.py all --mbases 3160 --reps 1 --simple-cov 1 > $R/gpurun_out/pmc_step.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_step/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.split(r"[<(]", r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::",""))[0][-40:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
rows=[]
for k,c in agg.items():
    d=max(1,len(n[k]))
    tot=sum(v for kk,v in c.items() if kk.startswith("SQ_INSTS"))
    rows.append((tot/d, k, d, {kk.replace("SQ_INSTS_",""): round(v/d/1e6,2) for kk,v in c.items()}))
for t,k,d,c in sorted(rows,reverse=True)[:16]: print("%9.1f M wave-instr/launch  %-40s x%d %s"%(t/1e6,k,d,c))
PY
rm -rf gpurun_out/pmc_step
