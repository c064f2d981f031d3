#!/usr/bin/env python3
"""Generate tests/golden/seq_ops_as_written.json: small sequences, op codes, and the packed words the REFERENCE's
gasal_reversecomplement_kernel (AGAThA/src/kernels/pack_rc_seqs.h:56-212) leaves behind, as restated line by line in
oracle/seq_ops_ref.c (agatha_ref_seq_ops_as_written; the header of that file says why the kernel itself is not run
under the CPU shim).  Data only: ASCII sequences, op codes, uint32 words.  Run from the repo root:
    python tests/golden/gen_seq_ops_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(0x5E9095)
seqs, ops = [], []
for length in list(range(1, 34)) + [40, 63, 64, 65, 127, 128, 200, 256, 1001]:
    for op in (1, 2, 3):
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, length)].copy()
        if length > 4 and rng.random() < 0.3:
            s[rng.integers(0, length)] = ord("N")
        seqs.append(s.tobytes().decode())
        ops.append(op)
buf, offs, lens = O.make_batch([s.encode() for s in seqs])
packed = O.pack(buf)
out = O.seq_ops(packed, lens, offs, np.asarray(ops, np.uint8), as_written=True)
doc = {"source": "oracle/seq_ops_ref.c: agatha_ref_seq_ops_as_written (restatement of pack_rc_seqs.h:56-212)",
       "seqs": seqs, "ops": ops, "packed_after": [int(v) for v in out]}
json.dump(doc, open(os.path.join(os.path.dirname(__file__), "seq_ops_as_written.json"), "w"))
print(len(seqs), "sequences,", out.size, "words")
