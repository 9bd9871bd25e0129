#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/$1 on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench line (config 4, bf16) and of --dtype f32 / c3 / c2 / the engine workload,
#   FETCH_SIZE and WRITE_SIZE counter passes (separate runs, kernel-trace only) for config 4 in both element types.
# tools/pmc_summary.py then turns the counter passes into profiles/pmc_c4[_bf16].json; copy the *_kernel_stats.csv and
# bench lines into profiles/ under the round's prefix.
set -o pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
run_stats() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- python3 "$B" "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "stats $name failed"
    echo "stats $name done"
}
run_pmc() {    # name, counter, bench args...
    local name=$1 counter=$2; shift 2
    rocprofv3 --kernel-trace --pmc "$counter" --output-format csv -d "$OUT/pmc_${counter}_$name" -- python3 "$B" "$@" > /dev/null 2> "$OUT/pmc_${counter}_$name.err" || echo "pmc $counter $name failed"
    echo "pmc $counter $name done"
}
run_stats c4_bf16 --no-configs --no-cpu-baseline
run_stats c4_f32 --dtype f32 --no-configs --no-cpu-baseline
run_stats c3 --workload c3 --no-configs --no-cpu-baseline
run_stats c2 --workload c2 --no-configs --no-cpu-baseline
run_stats engine_e1 --mode engine
run_pmc c4_bf16 FETCH_SIZE --steps 10 --warmup 2 --no-configs --no-cpu-baseline
run_pmc c4_bf16 WRITE_SIZE --steps 10 --warmup 2 --no-configs --no-cpu-baseline
run_pmc c4_f32 FETCH_SIZE --dtype f32 --steps 10 --warmup 2 --no-configs --no-cpu-baseline
run_pmc c4_f32 WRITE_SIZE --dtype f32 --steps 10 --warmup 2 --no-configs --no-cpu-baseline
echo all done
