cd /tmp && export TMPDIR=/tmp
for k in tridiag_ql tridiag_adj; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$k -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py --kernel $k --reps 3 --shapes 7:100:10000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$k.log 2>&1
done
