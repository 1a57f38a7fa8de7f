#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box:  bash profiles/collect.sh <tag> [part ...]
# Parts (default: all): bench  stats  pmc_sparse  stats_dense  pmc_dense  pmc_bnf  stats_bnf
# Everything lands in gpurun_out/<tag>_prof/; profiles/summarize.py turns it into the files kept under profiles/.
# gpurun merges what a run wrote into the local gpurun_out/ and deletes nothing there: remove the local
# gpurun_out/<tag>_prof before collecting a part again, or the summaries mix two builds.
# Counter passes run on their own (no tracing beside --pmc) and each within the per-block counter limits.
set -u
TAG=${1:?tag}; shift
PARTS=${*:-bench stats pmc_sparse stats_dense pmc_dense pmc_bnf stats_bnf}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

pmc() {   # pmc <name> <counters...> -- <program args...>
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    echo "[collect] pmc $name: ${ctrs[*]}"
    rm -rf "$OUT/pmc_$name"
    timeout -s KILL 240 rocprofv3 --pmc "${ctrs[@]}" -d "$OUT/pmc_$name" --output-format csv -- python3 "$@" > "$OUT/pmc_$name.log" 2>&1 || { echo "[collect] pmc $name failed"; return 1; }
}

for part in $PARTS; do
    case $part in
    bench)
        echo "[collect] bench.py"
        timeout -k 10 400 python3 "$ROOT/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || exit 1 ;;
    stats)
        echo "[collect] kernel stats, bench.py"
        rm -rf "$OUT/stats_bench"
        timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench" --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats_bench.log" 2>&1 || exit 1 ;;
    pmc_sparse)
        S="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
        pmc fetch FETCH_SIZE -- $S || exit 1
        pmc write WRITE_SIZE -- $S || exit 1
        pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $S || exit 1
        pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $S || exit 1 ;;
    stats_dense)
        for mp in lists dense csr prebuilt; do
            echo "[collect] kernel stats, bench_dense --mask-path $mp"
            rm -rf "$OUT/stats_dense_$mp"
            timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dense_$mp" --output-format csv -- python3 "$ROOT/benchmarks/bench_dense.py" --mask-path $mp > "$OUT/stats_dense_$mp.log" 2>&1 || exit 1
        done
        for mp in lists prebuilt; do          # the reference's training rates (execute_cora.py:107)
            echo "[collect] kernel stats, bench_dense --mask-path $mp --dropout 0.6"
            rm -rf "$OUT/stats_dense_${mp}_drop"
            timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dense_${mp}_drop" --output-format csv -- python3 "$ROOT/benchmarks/bench_dense.py" --mask-path $mp --dropout 0.6 > "$OUT/stats_dense_${mp}_drop.log" 2>&1 || exit 1
        done
        : > "$OUT/bench_dense.jsonl"
        for args in "--mask-path lists" "--mask-path prebuilt" "--mask-path lists --dropout 0.6" "--mask-path prebuilt --dropout 0.6" "--mask-path dense" "--mask-path csr"; do
            timeout -k 10 200 python3 "$ROOT/benchmarks/bench_dense.py" $args >> "$OUT/bench_dense.jsonl" 2>> "$OUT/bench_dense.err" || exit 1
        done
        timeout -k 10 300 python3 "$ROOT/benchmarks/bench_two_layer.py" > "$OUT/bench_two_layer.jsonl" 2> "$OUT/bench_two_layer.err" || exit 1
        timeout -k 10 300 python3 "$ROOT/benchmarks/bench_two_layer.py" --dropout 0.6 >> "$OUT/bench_two_layer.jsonl" 2>> "$OUT/bench_two_layer.err" || exit 1 ;;
    pmc_dense)
        for mp in lists dense prebuilt; do
            D="$ROOT/benchmarks/bench_dense.py --mask-path $mp --steps 2 --warmup 1"
            pmc dense_${mp}_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $D || exit 1
            pmc dense_${mp}_fetch FETCH_SIZE -- $D || exit 1
            pmc dense_${mp}_write WRITE_SIZE -- $D || exit 1
        done ;;
    pmc_bnf)
        pmc bnf_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- "$ROOT/benchmarks/bench_bnf.py" || exit 1 ;;
    stats_bnf)
        echo "[collect] bench_bnf.py, and kernel stats of the autoencoder step at the sizes of BNF_AE.py"
        timeout -k 10 200 python3 "$ROOT/benchmarks/bench_bnf.py" > "$OUT/bench_bnf.json" 2> "$OUT/bench_bnf.err" || exit 1
        AE="$ROOT/examples/train_bnf_ae_like.py --quiet --nodes 246 --hid 64 --heads 6 --cam 256"
        timeout -k 10 200 python3 $AE --steps 100 > "$OUT/bnf_autoencoder.txt" 2>> "$OUT/bench_bnf.err" || exit 1
        timeout -k 10 200 python3 $AE --steps 100 --replay >> "$OUT/bnf_autoencoder.txt" 2>> "$OUT/bench_bnf.err" || exit 1
        rm -rf "$OUT/stats_bnf_ae"
        timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bnf_ae" --output-format csv -- python3 $AE --steps 52 --replay > "$OUT/stats_bnf_ae.log" 2>&1 || exit 1
        find "$OUT/stats_bnf_ae" -name "*kernel_trace.csv" -delete ;;
    *) echo "unknown part $part"; exit 2 ;;
    esac
done
echo "[collect] done: $OUT"
