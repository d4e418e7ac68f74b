# rocprofv3 --pmc passes over tools/gemm_pmc_probe.py, summarised per GEMM kernel.   gpurun -- bash tools/gemm_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/pmc_gemm_a gpurun_out/pmc_gemm_b gpurun_out/pmc_gemm_c
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pmc_gemm_a -o pmc --output-format csv -- python3 tools/gemm_pmc_probe.py > gpurun_out/pmc_gemm_a.log 2>&1 || { echo "pass a failed"; tail -5 gpurun_out/pmc_gemm_a.log; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_gemm_b -o pmc --output-format csv -- python3 tools/gemm_pmc_probe.py > gpurun_out/pmc_gemm_b.log 2>&1 || { echo "pass b failed"; tail -5 gpurun_out/pmc_gemm_b.log; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d gpurun_out/pmc_gemm_c -o pmc --output-format csv -- python3 tools/gemm_pmc_probe.py > gpurun_out/pmc_gemm_c.log 2>&1 || { echo "pass c failed"; tail -5 gpurun_out/pmc_gemm_c.log; }
python3 - <<PY
import csv,collections,re,os
for d in ('pmc_gemm_a','pmc_gemm_b','pmc_gemm_c'):
    f='gpurun_out/%s/pmc_counter_collection.csv'%d
    if not os.path.exists(f):
        print(d,'no output'); continue
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in rows:
        n=r['Kernel_Name']
        m=re.search(r'(gemm_\w+_kernel(<[^>]*>)?)',n)
        if m:
            key=m.group(1)
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[key].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    for k,v in agg.items():
        print(d,k,"dur_us %.1f"%(sum(dur[k])/len(dur[k])/1e3),{c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
