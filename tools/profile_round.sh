#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/$1 on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench line (config 4, bf16) and of --dtype f32 / fp8 / c3 / c2 / the engine workload
#   (default loop; sequential loop with lean layers and with the reference's launch sequence),
#   FETCH_SIZE and WRITE_SIZE counter passes (separate runs, kernel-trace only) for config 4 in the three element types and
#   for configs 3 and 2.
# tools/pmc_summary.py then turns the counter passes into profiles/pmc_<workload>.json; copy the *_kernel_stats.csv and
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
    rocprofv3 --kernel-trace --pmc "$counter" --output-format csv -d "$OUT/pmc_${counter}_$name" -- python3 "$B" "$@" > "$OUT/pmc_${counter}_$name.json" 2> "$OUT/pmc_${counter}_$name.err" || echo "pmc $counter $name failed"
    echo "pmc $counter $name done"
}
run_stats c4_bf16 --no-configs --no-cpu-baseline
run_stats c4_f32 --dtype f32 --no-configs --no-cpu-baseline
run_stats c4_fp8 --dtype fp8 --no-configs --no-cpu-baseline
run_stats c3 --workload c3 --no-configs --no-cpu-baseline
run_stats c2 --workload c2 --no-configs --no-cpu-baseline
run_stats engine_e1 --mode engine
run_stats engine_e1_sequential_lean --mode engine --sequential-loop
run_stats engine_e1_sequential_reference --mode engine --sequential-loop --reference-launch-sequence
for cfg in "c4_bf16" "c4_f32 --dtype f32" "c4_fp8 --dtype fp8" "c3 --workload c3" "c2 --workload c2"; do
    set -- $cfg
    name=$1; shift
    run_pmc $name FETCH_SIZE "$@" --steps 10 --warmup 2 --repeats 1 --no-configs --no-cpu-baseline
    run_pmc $name WRITE_SIZE "$@" --steps 10 --warmup 2 --repeats 1 --no-configs --no-cpu-baseline
done
echo all done
