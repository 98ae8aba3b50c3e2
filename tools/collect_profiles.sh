#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root): bench lines, rocprofv3 kernel
# stats, PMC passes (FETCH / WRITE / SQ in separate runs, as MI355X_MICROARCH.md prescribes) and the
# binding / binary regimes.  Writes under gpurun_out/prof_$TAG; copy the summaries into profiles/.
# Two parts (a gpurun call is at most 20 minutes): A = bench lines, kernel stats, the headline sweep's PMC passes, the
# binding / binary regimes; B = the T = 96 sweeps (125 000 and 1 000 000 residences), the matrix-core product, the feeder,
# the bursts' trace.
TAG=${1:-r05}
PART=${2:-AB}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *A* ]]; then
python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_steps20.json 2> $O/bench20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-converge > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "sq2:SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/$name -o $name -- python3 $R/bench.py --steps 128 --warmup 32 --no-extras --no-cpu-baseline --no-converge --clock-warm 0 > $O/pmc_$name.log 2>&1
done
for reg in binding binary; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/reg_$reg -o s -- python3 $R/tools/regime_run.py --regime $reg > $O/reg_$reg.log 2>&1
  cp $(find $O/reg_$reg -name "*kernel_stats.csv" | head -1) $O/${reg}_kernel_stats.csv
  rm -rf $O/reg_$reg
done
find $O/pmc -type f ! -name "*counter_collection.csv" -delete
fi
if [[ $PART == *B* ]]; then
# BASELINE config 4's per-GPU shape: SQ counters and HBM traffic of the T = 96 sweep (roofline_125k_T96)
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_t96/$name -o $name -- python3 $R/tools/regime_run.py --regime steady --homes 125000 --T 96 --steps 128 --spin 48 > $O/pmc_t96_$name.log 2>&1
  python3 $R/tools/pmc_kernels.py $O/pmc_t96/$name agent_step > $O/pmc_t96_$name.txt 2>&1
done
find $O/pmc_t96 -type f ! -name "*summary.csv" -delete
# BASELINE config 4 at its whole size on one GPU (1 000 000 x 96): the same three passes (roofline_1M_T96) -- the residences'
# state (6.5 GB) no longer fits the Infinity Cache: the multi-iteration sweep's HBM figure
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_t96_1m/$name -o $name -- python3 $R/tools/regime_run.py --regime steady --homes 1000000 --T 96 --steps 64 --spin 48 > $O/pmc_t96_1m_$name.log 2>&1
  python3 $R/tools/pmc_kernels.py $O/pmc_t96_1m/$name agent_step > $O/pmc_t96_1m_$name.txt 2>&1
done
find $O/pmc_t96_1m -type f ! -name "*summary.csv" -delete
# the f64 matrix-core product R p: MFMA counters at the synthetic feeder's shape (M = 2048, T = 24) and at BASELINE config 3's own
# (the 121144 feeder's 1 126 residence rows, T = 96)
for shape in "syn:" "config3:--config3"; do
  sn=${shape%%:*}; sa=${shape#*:}
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_mfma_$sn -o mfma -- python3 $R/tools/matvec_run.py $sa > $O/pmc_mfma_$sn.log 2>&1
  python3 $R/tools/pmc_kernels.py $O/pmc_mfma_$sn gemm_tn > $O/pmc_mfma_$sn.txt 2>&1
  cp $O/pmc_mfma_$sn/pmc_kernels_summary.csv $O/pmc_mfma_$sn.csv
  rm -rf $O/pmc_mfma_$sn
done
# stage stamps of the folded chain's operator launch (tuning build, if present)
if [ -f $R/tune/librevs_stamps.so ]; then
  for reg in binding binary; do REVS_LIB=$R/tune/librevs_stamps.so python3 $R/tools/regime_run.py --regime $reg --steps 100 2>&1 | grep "kv stamps" > $O/kv_stamps_$reg.txt; done
fi
# the reference's own case: kernel stats of the 15-iteration runs on the 121144 feeder
rocprofv3 --kernel-trace --stats --output-format csv -d $O/feeder -o s -- python3 $R/tests/tools/feeder_iters.py > $O/feeder.log 2>&1
cp $(find $O/feeder -name "*kernel_stats.csv" | head -1) $O/feeder_kernel_stats.csv
rm -rf $O/feeder
if [ -f $R/tune/librevs_bpp.so ]; then REVS_LIB=$R/tune/librevs_bpp.so python3 $R/tools/bpp_stamps.py 2>&1 | tail -4 > $O/bpp_stamps.txt; fi
# the verdict launch behind a burst of 20 / a block of 32 iterations (tuning build), and the bursts' kernel trace
if [ -f $R/tune/librevs_vd.so ]; then for k in 20 32; do REVS_LIB=$R/tune/librevs_vd.so python3 $R/tools/verdict_stamps.py $k 2>&1 | tail -7 > $O/verdict_stamps_$k.txt; done; fi
rocprofv3 --kernel-trace --output-format csv -d $O/tr20 -o t -- python3 $R/bench.py --steps 20 --no-extras --no-cpu-baseline --no-converge > $O/tr20.log 2>&1
python3 $R/tools/burst_trace.py $O/tr20 > $O/burst_trace_steps20.txt 2>&1
rm -rf $O/tr20
rm -rf $O/kt
fi
ls -la $O
