# Two rocprofv3 --pmc passes (issue / wait counters, then LDS / MFMA / VMEM counters) over the conv layer 0 kernels
# (tools/bench_kernels.py conv0), summarised per kernel.   gpurun -- bash tools/conv0_pmc.sh [W2VS_CONV0_BWD form]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/pmc_conv0_a gpurun_out/pmc_conv0_b
if [ -n "$1" ]; then export W2VS_LIB=$R/wav2vec-s_amd/libw2vs_tuning.so W2VS_CONV0_BWD=$1; fi
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pmc_conv0_a -o pmc --output-format csv -- python3 tools/bench_kernels.py conv0 > gpurun_out/pmc_conv0_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d gpurun_out/pmc_conv0_b -o pmc --output-format csv -- python3 tools/bench_kernels.py conv0 > gpurun_out/pmc_conv0_b.log 2>&1
python3 - <<PY
import csv,collections,re
for d in ('pmc_conv0_a','pmc_conv0_b'):
    rows=list(csv.DictReader(open('gpurun_out/%s/pmc_counter_collection.csv'%d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in rows:
        m=re.search(r'(conv0_mfma_\w+_kernel)',r['Kernel_Name'])
        if m:
            key=m.group(1)
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[key].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    for k,v in agg.items():
        print(d,"form=${1:-product}",k,"dur_us %.1f"%(sum(dur[k])/len(dur[k])/1e3),{c:"%.3g"%(sum(x)/len(x)) for c,x in v.items()})
PY
