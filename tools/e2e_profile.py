import cProfile, pstats, sys, tempfile
sys.path.insert(0, ".")
from strkit_amd.frontend import Fasta, call_sample, read_bam
from strkit_amd.frontend.synth_dataset import make_dataset
d = tempfile.mkdtemp()
t = make_dataset(d, n_loci=300, reads_per_locus=30, read_len=3000, seed=1, sub=0.001, indel=0.002)
bam, ref = read_bam(t["paths"]["bam"]), Fasta(t["paths"]["ref"])
call_sample(bam, ref, t["paths"]["loci"])
pr = cProfile.Profile(); pr.enable(); call_sample(bam, ref, t["paths"]["loci"]); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
