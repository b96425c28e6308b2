#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:
#     scripts/profile_gpu.sh <tag> [workload ...]        workloads: headline configs1 configs3 configs4 fit dense
# For every workload: rocprofv3 --kernel-trace --stats of `bench.py --only <workload>`, then separate PMC passes
# (HBM fetch / write bytes, SQ issue counters, LDS counters) -- never combined with a trace domain.  The program
# itself follows `--` (python3 bench.py ...).  Outputs under gpurun_out/<tag>/<workload>/; condense with
# scripts/summarize_profile.py gpurun_out/<tag> profiles/<round>.
set -o pipefail
TAG=${1:-prof}
shift
WORKLOADS=${@:-headline configs1 configs3 configs4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  OUT=$ROOT/gpurun_out/$TAG/$W
  mkdir -p $OUT
  BENCH="python3 $ROOT/bench.py --only $W --steps 3 --warmup 1"
  echo "== $W: plain run (steps per pass)"
  $BENCH > $OUT/plain.json 2> $OUT/plain.err || { tail -20 $OUT/plain.err; exit 1; }
  echo "== $W: kernel trace" && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
  echo "== $W: pmc FETCH_SIZE" && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
  echo "== $W: pmc WRITE_SIZE" && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
  echo "== $W: pmc SQ" && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; exit 1; }
  echo "== $W: pmc SQ (LDS / waits)" && rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || { tail -20 $OUT/pmc_sq2.log; exit 1; }
done
echo "profile done: $WORKLOADS"
