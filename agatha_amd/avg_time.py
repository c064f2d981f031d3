"""Average kernel time per iteration from a raw log -- the post-processing step of the reference's driver script
(reference misc/avg_time.py:14-44, called by AGAThA.sh:60).

    python -m agatha_amd.avg_time PROCESS DATASET_ID RAW_FILE OUTPUT_JSON ITERATIONS

RAW_FILE holds one float (milliseconds) per line, appended by `manual -p ... raw.log` for every batch of every
iteration.  OUTPUT_JSON gets {PROCESS: {DATASET_ID: sum / ITERATIONS}} merged into whatever it already holds; the value
is the string "NaN" when the raw file is missing or empty, as in the reference.
"""
import json
import os
import sys


def average(raw_file, iterations):
    if not os.path.exists(raw_file):
        return "NaN"
    with open(raw_file) as f:
        lines = f.read().splitlines()
    if not lines:
        return "NaN"
    return sum(float(x) for x in lines) / float(iterations)


def update(process, dataset_id, raw_file, output_file, iterations):
    out = {}
    if os.path.exists(output_file):
        with open(output_file) as f:
            out = json.load(f)
    out.setdefault(process, {})[dataset_id] = average(raw_file, iterations)
    with open(output_file, "w") as f:
        json.dump(out, f)
    return out


def main(argv=None):
    a = sys.argv[1:] if argv is None else argv
    if len(a) != 5:
        sys.stderr.write(__doc__)
        return 2
    update(a[0], a[1], a[2], a[3], int(a[4]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
