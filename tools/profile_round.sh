#!/bin/bash
# Produce the round's profile artefacts on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh OUTDIR [LIBTAG]
# 1. rocprofv3 --kernel-trace --stats of `bench.py --steps 30 --warmup 2 --no-extras` (the summary averages ALL launches, the two
#    warm-up steps included: 30 timed steps keep that within 1 % of the timed average the bench prints) -> OUTDIR/stats/
# 2. two PMC passes (FETCH_SIZE / WRITE_SIZE, separate: TCC slots) over a short bench -> OUTDIR/pmc_fetch, pmc_write
# 3. two SQ passes (8 counters each)                                          -> OUTDIR/pmc_sq1, pmc_sq2
# Counter runs carry --kernel-trace only (gpurun refuses --pmc beside the hip/hsa trace domains).
out=$1; tag=$2
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$tag" ]; then export EMAVFI_LIB=$PWD/video-frame-interpolation_amd/emavfi/lib/libemavfi_$tag.so; fi
short="python3 bench.py --steps 2 --warmup 1 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 30 --warmup 2 --no-extras > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $short > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $short > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq1 -- $short > $out/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- $short > $out/pmc_sq2.log 2>&1 || exit 1
# keep only the small CSVs (the merge back is limited to 64 MiB)
find $out -name "*.db" -delete
find $out -name "*_kernel_trace.csv" -size +8M -delete
ls -R $out | head -40
