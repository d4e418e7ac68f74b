cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_a1 $R/gpurun_out/pmc_a2
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pmc_a1 -o pmc --output-format csv -- python tools/attn_probe.py one > gpurun_out/pmc_a1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d gpurun_out/pmc_a2 -o pmc --output-format csv -- python tools/attn_probe.py one > gpurun_out/pmc_a2.log 2>&1
ls gpurun_out/pmc_a1 gpurun_out/pmc_a2; tail -3 gpurun_out/pmc_a2.log
