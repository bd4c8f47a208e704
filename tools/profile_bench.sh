# Profile the bench command on the GPU box (run through gpurun from the repo root): per-kernel times, then PMC counters in
# their own passes (never combined with trace domains).  Two commands: the default bench step (encoder + decoder kernels) and
# the joint-lattice leg (--legs joint), so every roofline object of the bench line has a `traffic` figure.
#   usage: bash tools/profile_bench.sh <tag> ["step joint"]      -> gpurun_out/prof_<tag>/ ; summaries via tools/{pmc,mfma}_summary.py
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${1:-x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for leg in ${2:-step joint}; do
  if [ $leg = step ]; then CMD="python3 $R/bench.py --no-legs --no-cpu --numerics bf16x3 --steps 10"; else CMD="python3 $R/bench.py --legs joint --no-cpu --numerics bf16x3 --steps 2 --in-flight 1"; fi
  rocprofv3 --kernel-trace --stats -d $O/kt_$leg -o kt -- $CMD > $O/bench_kt_$leg.json 2> $O/kt_$leg.err
  echo kt $leg done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/mfma_$leg -o mfma -- $CMD > $O/bench_mfma_$leg.json 2> $O/mfma_$leg.err
  echo mfma $leg done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$leg -o fetch -- $CMD > $O/bench_fetch_$leg.json 2> $O/fetch_$leg.err
  echo fetch $leg done
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$leg -o write -- $CMD > $O/bench_write_$leg.json 2> $O/write_$leg.err
  echo write $leg done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/lds_$leg -o lds -- $CMD > $O/bench_lds_$leg.json 2> $O/lds_$leg.err
  echo lds $leg done
done
find $O -name "*.csv" | head -40
