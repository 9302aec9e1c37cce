export TMPDIR=/tmp
for l in libcomposer_hip ad5 ad3 ad2; do
  out=gpurun_out/clk/$l; mkdir -p $out
  KB_B=128 COMPOSER_HIP_LIB=composer_amd/lib/$l.so timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out -o k -- python3 tools/kbench.py attn > $out.log 2>&1
  python3 - <<PY
import csv,glob,collections
d="$out"
dur=collections.defaultdict(list)
for f in glob.glob(d+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"][-30:])
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"] and r["Dispatch_Id"] in dur:
            acc[r["Kernel_Name"][-30:]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]][0]))
for k,cs in acc.items():
    g=cs["GRBM_GUI_ACTIVE"]; n=len(g)
    clk=sum(v/8/d_ for v,d_ in g)/n
    du=sum(d_ for v,d_ in g)/n/1e3
    print("$l",k,"dur %.1f us clock %.2f GHz"%(du,clk), {c: "%.3g"%(sum(v for v,_ in vs)/len(vs)) for c,vs in cs.items() if c!="GRBM_GUI_ACTIVE"})
PY
done
