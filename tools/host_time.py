import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
from strkit_amd import _lib
from strkit_amd.batch import make_params
from strkit_amd.synth import make_config
dev = torch.device("cuda", 0)
b = make_config(2)
L = _lib.load()
t = dict(seqs=torch.from_numpy(b.seqs).to(dev), seq_off=torch.from_numpy(b.seq_off).to(dev), nfl=torch.from_numpy(b.nfl).to(dev), ntr=torch.from_numpy(b.ntr).to(dev), nfr=torch.from_numpy(b.nfr).to(dev), est_cn=torch.from_numpy(b.est_cn).to(dev), read_off=torch.from_numpy(b.read_off).to(dev), motifs=torch.from_numpy(b.motifs).to(dev), motif_off=torch.from_numpy(b.motif_off).to(dev))
sb = _lib.StrkBatch(n_reads=b.n_reads, n_loci=b.n_loci, **{k: v.data_ptr() for k, v in t.items()})
p = make_params(); st = _lib.StrkStats()
D = 4
ctxs = [_lib.Context(0) for _ in range(D)]; streams = [torch.cuda.Stream(dev) for _ in range(D)]
outs = [torch.zeros((4, b.n_reads), dtype=torch.int32, device=dev) for _ in range(D)]
def submit(i):
    k = i % D; o = outs[k]
    L.strk_submit_loci_device(ctxs[k].handle, C.byref(sb), C.byref(p), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), C.c_void_p(streams[k].cuda_stream))
def finish(i):
    L.strk_finish(ctxs[i % D].handle, C.byref(st))
for i in range(D): submit(i)
for i in range(20): finish(i); submit(i + D)
ts = tf = 0.0; N = 100
t0 = time.perf_counter()
for i in range(20, 20 + N):
    a = time.perf_counter(); finish(i); b1 = time.perf_counter(); submit(i + D); c = time.perf_counter()
    tf += b1 - a; ts += c - b1
tot = time.perf_counter() - t0
for i in range(20 + N, 20 + N + D): finish(i)
print(f"per step: total {tot/N*1e3:.3f} ms, finish (incl. waiting) {tf/N*1e3:.3f} ms, submit {ts/N*1e3:.3f} ms")
