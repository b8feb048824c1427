#!/bin/bash
# Everything the profiles/ summaries are made from, in one GPU call: the kernel trace of the default bench
# run, the counter passes (one process each) and the full bench line.  Run through gpurun from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_* gpurun_out/prof_bench   # on the GPU box; delete the local copies too before the call
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --no-h2d --no-other-configs > gpurun_out/prof_bench.log 2>&1
tools/pmc_pass.sh a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > /dev/null
tools/pmc_pass.sh b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS > /dev/null
tools/pmc_pass.sh f FETCH_SIZE > /dev/null
tools/pmc_pass.sh w WRITE_SIZE > /dev/null
tools/pmc_pass.sh t TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum > /dev/null
tools/pmc_pass.sh m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE > /dev/null
# kernel trace of continuous-training iterations (config 4) and the host-phase breakdown of the same loop
rm -rf gpurun_out/prof_train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o train -- python3 tools/prof_train.py 10 > gpurun_out/prof_train_traced.log 2>&1
timeout -k 10 300 python3 tools/prof_train.py 20 > gpurun_out/prof_train.txt 2>&1
timeout -k 10 500 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
tail -c 600 gpurun_out/bench_line.json
