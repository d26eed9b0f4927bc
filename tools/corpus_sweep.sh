#!/bin/bash
# tools/corpus_sweep.sh <out dir> [max nnz per file] [chunk budget in nonzeros] -- the reference's corpus sweep (eval_csrmv.sh:8-17:
# `<driver> --quiet --mtx=<file>` over a directory of Matrix Market files, one CSV line per file) over the corpus-shaped stand-ins
# of tools/make_corpus.py, fp64 (the reference's default, gpu_spmv.cu:727-735) and fp32, chunk by chunk so that the scratch
# directory never holds more than one chunk of files.  Point MSPMV_CORPUS_DIR at a directory of REAL SuiteSparse files and the same
# eval_csrmv.sh lines run those instead (nothing is generated then).
#   <out>/corpus_fp64.csv, corpus_fp32.csv   the reference's CSV (tools/eval_csrmv.sh: header + one line per file; methods: ours, rocSPARSE csrmv)
#   <out>/corpus_checks.txt                  one `strict-check, file, precision, PASS|FAIL, violations, worst ratio` line per file and precision
#   <out>/corpus_files.txt                   what was written (size, entries, seconds)
set -u
OUT=${1:?out dir}; MAXNNZ=${2:-2.0e8}; BUDGET=${3:-2.5e8}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
SCRATCH=${MSPMV_CORPUS_SCRATCH:-/tmp/mspmv_corpus}
: > "$OUT/corpus_checks.txt"; : > "$OUT/corpus_files.txt"
df -h /tmp /dev/shm 2>/dev/null | sed "s/^/# /" >> "$OUT/corpus_files.txt"
header=1
sweep() {   # $1 = directory of .mtx files
    for prec in fp64 fp32; do
        flag=""; [ $prec = fp32 ] && flag="--fp32"
        bash "$HERE/tools/eval_csrmv.sh" "$1" gpu_spmv --no-hyb --check $flag 2>> "$OUT/corpus_checks.txt" | { if [ $header = 1 ]; then cat; else tail -n +2; fi; } >> "$OUT/corpus_$prec.csv"
    done
    header=0
}
rm -f "$OUT/corpus_fp64.csv" "$OUT/corpus_fp32.csv"
if [ -n "${MSPMV_CORPUS_DIR:-}" ] && [ -d "$MSPMV_CORPUS_DIR" ]; then
    echo "sweeping the files under $MSPMV_CORPUS_DIR" >&2
    sweep "$MSPMV_CORPUS_DIR"
else
    next=0
    total=$(python3 "$HERE/tools/make_corpus.py" --list --max-nnz "$MAXNNZ" | tail -1 | cut -d' ' -f1)
    while [ "$next" -lt "$total" ]; do
        rm -rf "$SCRATCH"; mkdir -p "$SCRATCH"
        got=$(python3 "$HERE/tools/make_corpus.py" --dir "$SCRATCH" --from "$next" --budget-nnz "$BUDGET" --max-nnz "$MAXNNZ" 2>> "$OUT/corpus_files.txt" | tail -1)
        case "$got" in "next "*) next=${got#next };; *) echo "make_corpus.py failed at file $next: $got" >&2; break;; esac
        sweep "$SCRATCH"
        echo "corpus: $next of $total files done ($(date +%T))" >&2
    done
    rm -rf "$SCRATCH"
fi
python3 "$HERE/tools/corpus_summary.py" "$OUT" > "$OUT/corpus_summary.txt"
tail -40 "$OUT/corpus_summary.txt"
