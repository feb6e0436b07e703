"""`python -m strkit_amd call <alignments.bam> --ref ref.fa --loci catalog.bed [--json out.json] [--realign]` — the
subset of `strkit call` (strkit/entry.py:20-342) that the device backend covers: per-read copy numbers per locus."""
from __future__ import annotations

import argparse
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="strkit_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("call", help="per-read repeat counts for every catalog locus")
    c.add_argument("read_file")
    c.add_argument("--ref", required=True)
    c.add_argument("--loci", required=True)
    c.add_argument("--json", default="-")
    c.add_argument("--vcf", default=None, help="also write the loci as VCF (read-level fields; alleles come from allele calling)")
    c.add_argument("--flank-size", type=int, default=70)
    c.add_argument("--min-avg-phred", type=int, default=13)
    c.add_argument("--max-reads", type=int, default=250)
    c.add_argument("--realign", action="store_true")
    c.add_argument("--front-end", choices=("auto", "device", "host"), default="auto",
                   help="where the alignment file is inflated, scanned and cut: on the GPU (whole below 24 GB: about six times the "
                        "file in device memory; larger files with a .bai in spans) or on the host cores, block by block through "
                        "the .bai; auto = device (a file of 24 GB or more needs its .bai for that)")
    c.add_argument("--span-mb", type=int, default=4096,
                   help="device front end, files of 24 GB and more: compressed megabytes of the file that go through device memory at a time")
    c.add_argument("--respect-ref", action="store_true")
    # same names as `strkit call` (strkit/entry.py:20-342); --seed is accepted for command-line compatibility (the
    # per-read path has no random component), --processes sizes the locus blocks as the reference does (loci.py:193)
    c.add_argument("--sample-id", default=None)
    c.add_argument("--processes", type=int, default=1)
    c.add_argument("--seed", type=int, default=None)
    c.add_argument("--rc-method", choices=("repalign",), default="repalign")
    c.add_argument("--max-rcn-iters", type=int, default=50)
    c.add_argument("--min-read-align-score", type=float, default=0.1)
    a = ap.parse_args(argv)
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:   # python -m torch.distributed.run --nproc-per-node N -m strkit_amd call ...: one rank per GPU
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("STRKIT_AMD_DEVICE", str(local))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from .frontend import call_sample, write_json
    from .repeat_count_params import RepeatCountParams
    rc = RepeatCountParams("repalign", a.max_rcn_iters, 3, 1)   # params.py:26-27,45
    rep = call_sample(a.read_file, a.ref, a.loci, flank_size=a.flank_size, realign=a.realign,
                      min_avg_phred=a.min_avg_phred, max_reads=a.max_reads, respect_ref=a.respect_ref,
                      sample_id=a.sample_id, processes=a.processes, rc_params=rc,
                      min_read_align_score=a.min_read_align_score, front_end=a.front_end, span_bytes=a.span_mb << 20)
    if world > 1:
        import torch.distributed as dist
        rank0 = dist.get_rank() == 0
        dist.destroy_process_group()
        if not rank0:
            return 0
    if a.vcf:
        from .frontend.fasta import Fasta
        from .frontend.output import write_vcf
        write_vcf(rep, a.vcf, Fasta(a.ref), a.sample_id)
    if a.json == "-":
        import json
        json.dump(rep, sys.stdout, indent=1)
    else:
        write_json(rep, a.json)
    return 0


if __name__ == "__main__":
    sys.exit(main())
