#!/bin/bash
# Regenerates the judged artifacts of a round on the GPU box (run from the repo root, through gpurun):
#   kernel summaries of the three benchmarked configurations and of the generator-only graph, the two PMC passes
#   of the cfg2 step body, the three bench lines and the MLP-tail tuning table.  Everything lands under
#   gpurun_out/refresh/ with the file names profiles/ uses; copy from there.  The per-dispatch trace CSVs are
#   deleted on the box (they are tens of MiB; gpurun_out travels back only below 64 MiB).
#
#     gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r03_final'
set -eo pipefail
TAG=${1:-r03_final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
export TMPDIR=/tmp

stats() {                     # stats <name> <program...>: one kernel-trace pass, keeps only the summary
    local name=$1; shift
    local d=/tmp/prof_$name
    rm -rf "$d"
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- "$@" > "$OUT/$name.log" 2>&1)
    cp "$(find "$d" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_${name}_graph_bf16_kernel_stats.csv"
    rm -rf "$d"
    echo "profiled $name"
}

for cfg in cfg2 cfg4 cfg5shard; do
    stats $cfg python3 "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-extra
done
stats generator_only python3 "$ROOT/tools/gonly_profile.py" cfg2 12

for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc_$ctr
    rm -rf "$d"
    (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d" -- \
        python3 "$ROOT/bench.py" --eager-body --steps 3 --warmup 1 --no-extra > "$OUT/pmc_$ctr.log" 2>&1)
    echo "counted $ctr"
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > "$OUT/${TAG%_final}_pmc_traffic.json"
rm -rf /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
# what the summaries describe: executions of the step per profiled process (3 while capturing + warm-up + steps) and the
# state of the sources -- bench.py reports figures read from profiles/ only while its own sources still match
python3 - "$OUT/${TAG}_meta.json" <<'PYEOF'
import json, sys
sys.path.insert(0, ".")
import bench
json.dump({"executions": 15, "bench_args": "--steps 10 --warmup 2 --no-extra", "source": bench.source_fingerprint(),
           "note": "3 executions while capturing + 2 warm-up + 10 timed = 15 per profiled process"}, open(sys.argv[1], "w"), indent=1)
PYEOF
# bench.py quotes the library GEMM time and the PMC traffic from profiles/: give it the files just measured
cp "$OUT"/${TAG}_*_kernel_stats.csv "$OUT/${TAG}_meta.json" "$OUT/${TAG%_final}_pmc_traffic.json" "$ROOT/profiles/"

python3 "$ROOT/bench.py" > "$OUT/${TAG%_final}_bench_cfg2_bf16.json" 2> "$OUT/bench_cfg2.err"
python3 "$ROOT/bench.py" --config cfg4 --steps 100 --warmup 5 --no-extra > "$OUT/${TAG%_final}_bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
python3 "$ROOT/bench.py" --config cfg5shard --steps 60 --warmup 5 --no-extra > "$OUT/${TAG%_final}_bench_cfg5shard.json" 2> "$OUT/bench_cfg5shard.err"
echo "bench lines done"
python3 "$ROOT/tools/tune_mlp.py" > "$OUT/${TAG%_final}_tune_mlp.txt" 2>&1
echo "refresh complete"
