set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-legs --no-cpu --numerics bf16x3 --steps 10"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/bench_kt.json 2> $O/kt.err
echo kt done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/mfma -o mfma -- $CMD > $O/bench_mfma.json 2> $O/mfma.err
echo mfma done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $CMD > $O/bench_fetch.json 2> $O/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $CMD > $O/bench_write.json 2> $O/write.err
echo write done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/lds -o lds -- $CMD > $O/bench_lds.json 2> $O/lds.err
echo lds done
ls -R $O | head -40
