# PMC passes over the scatter kernels (modes 2, 3, 4) on P4; csv outputs under gpurun_out/pmc_r2/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_r2
mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | grep -E "LDS|WAIT|BUSY|ACTIVE|VMEM|WAVE" > $O/avail.txt
for m in 2 3 4; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
    tag=$(echo $set | md5sum | cut -c1-6)
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/m${m}_$tag -o p -- python $R/tools/run_scatter.py P4 tile_w=4 tile_h=4 back_mode=$m > $O/m${m}_$tag.log 2>&1 || echo "pass failed: m$m $set"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O='/root/repo/gpurun_out/pmc_r2'
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/m*_*/**/*counter_collection.csv', recursive=True):
    m=os.path.relpath(f,O).split('_')[0]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'back_' not in k: continue
        name=k.split('(')[0].replace('void (anonymous namespace)::','')[:40]
        res[(m,name)][r['Counter_Name']].append(float(r['Counter_Value']))
for (m,name),d in sorted(res.items()):
    print(m,name)
    for c,v in sorted(d.items()):
        v=v[len(v)//2:]   # later launches (after tuning / warm-up)
        print('   %-28s %14.4g  (n=%d)'%(c,sum(v)/len(v),len(v)))
PY
