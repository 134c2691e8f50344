#!/bin/bash
# rocprofv3 evidence of one round (GPU box): kernel-trace stats of pinned-threshold bench steps + separate --pmc passes.
# usage: bash tools/profile_round.sh r02     -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_summary.csv
set -o pipefail
TAG=${1:-rXX}
THR=${THR:-0.524019,0.542159,0.507206,0.430526,0.883812}      # thresholds of the default bench (round 5: 2 x 1024 documents, release 0.2, calibrated on seed 501234)
OUT=$PWD/gpurun_out
ROOT=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# thresholds AND the exit-layer schedule pinned: every forward of every pass is the same launch sequence
PLAN=${PLAN:-1,3,5,7,9}
# --distinct-batches 1: every forward of every pass runs resident batch 0, the batch of the bench line's own HIP-event step (the roofline's launches), so that the
# per-kernel averages here are averages over IDENTICAL launches and must agree with the line's avg_launch_ms
# --serial-slices: the two micro-batches one after the other on one stream, so that every kernel's start-to-end time is its own (with two
# streams a small kernel's duration includes its wait for the other stream's kernel and the per-kernel times stop adding up to the step)
COMMON="--cpu-docs 0 --stream-docs 0 --no-traffic --no-profile --serial-slices --distinct-batches 1 --no-small-batch --thresholds $THR --probe-layers $PLAN"
rm -rf $OUT/${TAG}_stats && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o t -- python3 $ROOT/bench.py --steps 8 --warmup 1 $COMMON > $OUT/${TAG}_stats.log 2>&1
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_UNALIGNED_STALL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmc$i
  rocprofv3 --pmc $set --output-format csv -d $OUT/${TAG}_pmc$i -o t -- python3 $ROOT/bench.py --steps 1 --warmup 0 $COMMON > $OUT/${TAG}_pmc$i.log 2>&1 || echo "pmc pass $i failed ($set)"
done
python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.csv $OUT/${TAG}_pmc1 $OUT/${TAG}_pmc2 $OUT/${TAG}_pmc3 $OUT/${TAG}_pmc4 $OUT/${TAG}_pmc5 $OUT/${TAG}_pmc6 $OUT/${TAG}_pmc7 $OUT/${TAG}_pmc8
# keep only the summaries (the raw per-dispatch CSVs are tens of MB)
rm -rf $OUT/${TAG}_pmc[0-9] $OUT/${TAG}_stats
