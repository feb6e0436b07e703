#!/bin/bash
# The inflater under AddressSanitizer + UBSan on the host (no GPU needed): a synthetic BAM and streams that exercise every copy
# path (runs of every short period, matches of every length at every kind of distance, all block types and strategies).
set -e
D=${TMPDIR:-/tmp}/strk_inflate_asan
mkdir -p $D
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o $D/inflate_asan tools/inflate_asan.cpp -lz
python3 - "$D" <<'PY'
import sys, zlib
import numpy as np
sys.path.insert(0, ".")
from strkit_amd.frontend.bam import bgzf_block
from strkit_amd.frontend.synth_dataset import make_dataset
d = sys.argv[1]
make_dataset(d, n_loci=40, reads_per_locus=12, read_len=4000, seed=21, sub=0.01, indel=0.01)
rng = np.random.default_rng(4)
payloads = [b"A", bytes(rng.integers(256, size=40000, dtype=np.uint8)), b"ACGT" * 9000, bytes(60000), bytes(rng.integers(33, 74, size=65000, dtype=np.uint8))]
runs = bytearray()
for period in range(1, 41):
    pat = bytes(rng.integers(256, size=period, dtype=np.uint8))
    for reps in (3, 11, 40, 300 // period + 2):
        runs += pat * reps + bytes(rng.integers(256, size=int(rng.integers(1, 9)), dtype=np.uint8))
far = bytearray(bytes(rng.integers(256, size=3000, dtype=np.uint8)))
for ln in list(range(3, 40)) + [63, 64, 65, 127, 128, 129, 130, 200, 257, 258, 259, 300, 600]:
    for back in (ln, ln + 1, 31, 32, 33, 127, 128, 129, 263, 264, 265, 2000):
        if 1 <= back <= len(far):
            src = len(far) - back
            far += bytes(far[src + i % back] for i in range(ln))
            far += bytes(rng.integers(256, size=int(rng.integers(0, 4)), dtype=np.uint8))
payloads += [bytes(runs[:65000]), bytes(far[:65000]), bytes(far[:60000]) + b"\x07" * 300, b"ab" * 150 + b"xyz" * 100]
for cut in (1, 2, 7, 8, 9, 31, 33, 127, 129, 263, 265):                       # matches that end exactly at the end of the block
    payloads.append(bytes(far[:50000]) + bytes(far[1000:1000 + 300]) [:300 - cut] )
out = bytearray()
for raw in payloads:
    for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_HUFFMAN_ONLY), (4, zlib.Z_RLE), (9, zlib.Z_DEFAULT_STRATEGY)):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = co.compress(raw) + co.flush()
        if len(body) + 26 <= 65536:
            out += bgzf_block(raw, body)
open(d + "/streams.bgzf", "wb").write(out)
PY
$D/inflate_asan $D/reads.bam $D/streams.bgzf
