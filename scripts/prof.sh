#!/bin/bash
# usage (GPU box): scripts/prof.sh NAME [bench args]  -> gpurun_out/NAME_{stats.csv,pmc.txt}
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (whole-frame batches from the first call on: the profiles are of the steady-state launches, not of a cold start's 16 M-sample batches)
export FJGPU_COLD_START=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name.stats -- python $root/bench.py --steps 2 --warmup 1 --cpu-tiles 0 "$@" > $out/$name.bench.json 2>$out/$name.err
# (the bench starts the gather calibration tool as a child, which rocprofv3 traces too: the bench's own file is the one with the walks in it)
for f in $(find $out/$name.stats -name "*kernel_stats.csv"); do grep -q 'k_trace_closest\|k_shadow' $f && cp $f $out/${name}_kernel_stats.csv; done
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$name.pmc_$tag -- python $root/bench.py --steps 1 --warmup 0 --cpu-tiles 0 "$@" > /dev/null 2>>$out/$name.err
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$out/$name.pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        if k.startswith('k_') or 'k_' in k: agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
import json
launches=collections.defaultdict(int)
for f in glob.glob("$out/$name.pmc_FETCH_SIZE/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        if r['Counter_Name']=='FETCH_SIZE' and 'k_' in k: launches[k]+=1
json.dump({"note": "one frame (bench.py --steps 1 --warmup 0) plus the untimed counting frame; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, collected in separate --pmc passes; gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section)",
           "launches": launches, "counters": {k: dict(v) for k, v in agg.items()}}, open("$out/${name}_pmc.json","w"), indent=1)
with open("$out/${name}_pmc.txt","w") as o:
    for k,v in sorted(agg.items()):
        o.write(k+"\n")
        for c,x in sorted(v.items()): o.write("   %-32s %.6g\n"%(c,x))
PY
cat $out/${name}_kernel_stats.csv | cut -c1-60,60-200 | head -7
cat $out/${name}_pmc.txt
