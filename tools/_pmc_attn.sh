cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ATT_ONLY40=1 BC_ATTN_QB=1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_a -o a --output-format csv -- python3 tools/attn_probe.py > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_b -o b --output-format csv -- python3 tools/attn_probe.py > gpurun_out/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY -d gpurun_out/pmc_c -o c --output-format csv -- python3 tools/attn_probe.py > gpurun_out/pmc_c.log 2>&1
ls -R gpurun_out/pmc_a | head; tail -3 gpurun_out/pmc_a.log
