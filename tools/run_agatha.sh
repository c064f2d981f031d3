#!/bin/bash
# Driver in the manner of the reference's AGAThA.sh (AGAThA.sh:1-62): runs the CLI ITER times on ref.fasta / query.fasta,
# collects the per-batch kernel times in raw.log, the scores in score.log and the average time per iteration in time.json.
#   tools/run_agatha.sh [-i ITER] [-d DATASET_DIR] [-o OUTPUT_DIR] [-n DATASET_NAME] [-- extra CLI flags]
# Defaults reproduce the reference's invocation: -m 1 -x 4 -q 6 -r 2 -s 3 -z 400 -w 751.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ITER=1; DATASET_DIR="$ROOT/dataset"; OUTPUT_DIR="$ROOT/output"; DATASET_NAME="test"; PROCESS="AGAThA"
while getopts "i:d:o:n:" opt; do
    case "$opt" in
    i) ITER="$OPTARG" ;; d) DATASET_DIR="$OPTARG" ;; o) OUTPUT_DIR="$OPTARG" ;; n) DATASET_NAME="$OPTARG" ;;
    esac
done
shift $((OPTIND - 1)); [ "$1" = "--" ] && shift
FLAGS=${*:-"-m 1 -x 4 -q 6 -r 2 -s 3 -z 400 -w 751"}
RAW_FILE="$OUTPUT_DIR/raw.log"; FINAL_FILE="$OUTPUT_DIR/time.json"; SCORE_FILE="$OUTPUT_DIR/score.log"
mkdir -p "$OUTPUT_DIR"
rm -f "$RAW_FILE" "$SCORE_FILE" "$FINAL_FILE"
echo ">>> Running $PROCESS for $ITER iterations."
for ((it = 1; it <= ITER; it++)); do
    echo ">> Iteration $it"
    "$ROOT/agatha_amd/manual" -p $FLAGS "$DATASET_DIR/ref.fasta" "$DATASET_DIR/query.fasta" "$RAW_FILE" > "$SCORE_FILE" || exit 1
done
echo "$PROCESS complete."
echo "Creating output files..."
python3 -m agatha_amd.avg_time "$PROCESS" "$DATASET_NAME" "$RAW_FILE" "$FINAL_FILE" "$ITER" || exit 1
echo "Complete."
