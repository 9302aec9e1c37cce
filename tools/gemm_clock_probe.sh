#!/bin/bash
# Effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of the 256x256 GEMM kernel: product build against the -DGEMM_DIAG
# measurement builds (composer_amd/lib/diag{0,1,4}.so from tools/ab_build.py), `kbench.py gemmdiag` shapes.
export TMPDIR=/tmp
for l in diag0 diag1 diag4; do
  out=gpurun_out/clk/$l; mkdir -p $out
  COMPOSER_HIP_LIB=composer_amd/lib/$l.so timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -o k -- python3 tools/kbench.py gemmdiag > $out.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for l in ('diag0','diag1','diag4'):
    # the three shapes run 23 launches each in order; split by order instead of grid
    allc=[]; alld=[]
    for f in glob.glob('gpurun_out/clk/%s/**/*counter_collection.csv'%l, recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if 'gemm_bf16_256' in r['Kernel_Name']]
        allc=[(int(r['Dispatch_Id']),float(r['Counter_Value'])) for r in rows]
    for f in glob.glob('gpurun_out/clk/%s/**/*kernel_trace.csv'%l, recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if 'gemm_bf16_256' in r['Kernel_Name']]
        alld=[(int(r['Dispatch_Id']),int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows]
    d=dict(alld); 
    seq=sorted(allc)
    n=len(seq)//3
    for i,name in enumerate(('K=512','K=2048','K=8192')):
        part=seq[i*n+3:(i+1)*n]
        clk=[c/8.0/d[k]  for k,c in part if k in d]
        us=[d[k]/1000.0 for k,c in part if k in d]
        if clk: print(l,name,'avg us %.1f'%(sum(us)/len(us)),'effective clock GHz %.3f'%(sum(clk)/len(clk)))
PY
find gpurun_out/clk -name "*.db" -delete
