# Profile the bench command on the GPU box (run through gpurun from the repo root): per-kernel times, then PMC counters in
# their own passes (never combined with trace domains); every pass under its own `timeout` (a pass that hangs ends the script: set -e).  Two commands: the default bench step (encoder + decoder kernels) and
# the joint-lattice leg (--legs joint), so every roofline object of the bench line has a `traffic` figure.
#   usage: bash tools/profile_bench.sh <tag> ["step joint"]      -> gpurun_out/prof_<tag>/ ; summaries via tools/{pmc,mfma}_summary.py
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${1:-x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for leg in ${2:-step joint}; do
  if [ $leg = step ]; then CMD="python3 $R/bench.py --no-legs --no-cpu --numerics bf16x3 --steps 10"; else CMD="python3 $R/bench.py --legs joint --no-cpu --numerics bf16x3 --steps 2 --in-flight 1"; fi
  echo kt $leg start; timeout -k 10 ${PASS_LIMIT:-150} rocprofv3 --kernel-trace --stats -d $O/kt_$leg -o kt -- $CMD > $O/bench_kt_$leg.json 2> $O/kt_$leg.err
  echo kt $leg done
  timeout -k 10 ${PASS_LIMIT:-150} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/mfma_$leg -o mfma -- $CMD > $O/bench_mfma_$leg.json 2> $O/mfma_$leg.err
  echo mfma $leg done
  timeout -k 10 ${PASS_LIMIT:-150} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$leg -o fetch -- $CMD > $O/bench_fetch_$leg.json 2> $O/fetch_$leg.err
  echo fetch $leg done
  timeout -k 10 ${PASS_LIMIT:-150} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$leg -o write -- $CMD > $O/bench_write_$leg.json 2> $O/write_$leg.err
  echo write $leg done
  timeout -k 10 ${PASS_LIMIT:-150} rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/lds_$leg -o lds -- $CMD > $O/bench_lds_$leg.json 2> $O/lds_$leg.err
  echo lds $leg done
done
find $O -name "*.csv" | head -40
# Summaries (the raw outputs exceed what gpurun copies back): gpurun_out/prof_<tag>_sum/, then the raw directories are removed.
S=$R/gpurun_out/prof_${1:-x}_sum
mkdir -p $S
for leg in ${2:-step joint}; do
  DB=$(find $O/kt_$leg -name "*.db" | head -1)
  [ -n "$DB" ] && python3 $R/tools/rocpd_stats.py $DB > $S/${leg}_kernel_stats.csv
  cp $O/bench_kt_$leg.json $S/${leg}_bench_under_kernel_trace.json
  python3 $R/tools/mfma_summary.py $(find $O/mfma_$leg -name "*counter_collection.csv" | head -1) $S/${leg}_mfma_busy.json > /dev/null
  python3 $R/tools/pmc_summary.py $(find $O/fetch_$leg -name "*counter_collection.csv" | head -1) $(find $O/write_$leg -name "*counter_collection.csv" | head -1) $S/${leg}_pmc_traffic.json > /dev/null
  python3 $R/tools/lds_summary.py $(find $O/lds_$leg -name "*counter_collection.csv" | head -1) $S/${leg}_lds_valu.json
  tail -3 $O/kt_$leg.err > $S/${leg}_kt_err_tail.txt
done
rm -rf $O
ls -la $S
