#!/usr/bin/env python3
"""Writes tests/golden/pathogenic_assoc.hg38.bed: the coordinate and ID/motif columns of the reference's own catalog
of disease-associated loci (catalogs/pathogenic_assoc.hg38.tsv, 44 loci, several IUPAC motifs such as AARRG, GCN,
RAAAT).  The file is data (a catalog), used to shape BASELINE.json's config 1; run from the repo root where
/root/reference exists:  python tests/golden/make_pathogenic_bed.py"""
import os

SRC = "/root/reference/catalogs/pathogenic_assoc.hg38.tsv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pathogenic_assoc.hg38.bed")

with open(SRC) as fh, open(DST, "w") as out:
    for line in fh:
        if line.startswith("#") or not line.strip():
            continue
        cols = line.rstrip("\n").split("\t")
        out.write("\t".join((cols[0], cols[1], cols[2], cols[-1])) + "\n")
print(DST)
