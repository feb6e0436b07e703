#!/usr/bin/env python3
"""Experiment harness (GPU box): for every library variant under strkit_amd/lib/exp/ (tools/exp_build.sh) measure
(a) the un-overlapped duration of k_dp_band / k_dp_all on one 10 000-locus batch of the bench workload, results checked
against the product library's on the same batch, and (b) the pipelined bench rate.
usage: python tools/exp_band.py [--config 2] [--bench] [names...]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, ctypes as C, numpy as np, hashlib
sys.path.insert(0, ".")
from strkit_amd import _lib
from strkit_amd.batch import batch_struct, make_params
from strkit_amd.synth import make_config, LocusBatch
cfg = int(sys.argv[1]); nb = int(sys.argv[2])
n_loci = {4: 21250, 5: 250}.get(cfg)
b = LocusBatch.concat([make_config(cfg, n_loci=n_loci, seed_shift=k) for k in range(nb)])
L = _lib.load(); ctx = _lib.default_context(0)
s, keep = batch_struct(b); p = make_params(); st = _lib.StrkStats()
outs = [np.zeros(b.n_reads, np.int32) for _ in range(4)]
tb, td, tk = [], [], []
tw = []
for i in range(16):
    L.strk_count_loci(ctx.handle, C.byref(s), C.byref(p), *[o.ctypes.data for o in outs], C.byref(st))
    if i >= 12: tb.append(st.band_kernel_ms); td.append(st.dp_kernel_ms); tk.append(st.kernel_ms); tw.append(st.band_wide_kernel_ms)
h = hashlib.sha1(b"".join(o.tobytes() for o in outs)).hexdigest()[:12]
print("band %.4f ms  exact %.4f ms  all %.4f ms  (head %.3f wide %.3f long %.3f generic %.3f replay %.3f)  band_reads %d fb %d win %d  cells %.3g  sha %s" % (sum(tb)/len(tb), sum(td)/len(td), sum(tk)/len(tk), st.head_ms, sum(tw)/len(tw), st.long_kernel_ms, st.generic_kernel_ms, st.replay_ms, st.n_band_reads, st.n_band_fallback, st.window_used, st.dp_cells, h))
'''


def main():
    args = sys.argv[1:]
    cfg = 2
    bench = False
    if "--config" in args:
        i = args.index("--config"); cfg = int(args[i + 1]); del args[i:i + 2]
    if "--bench" in args:
        bench = True; args.remove("--bench")
    dbgs = []          # extra runs of the product library with STRKIT_AMD_DBG set (e.g. 32: no staircase fork rows)
    while "--dbg" in args:
        i = args.index("--dbg"); dbgs.append(args[i + 1]); del args[i:i + 2]
    libs = {"product": os.path.join(ROOT, "strkit_amd", "lib", "libstrkit_amd.so")}
    for p in sorted(glob.glob(os.path.join(ROOT, "strkit_amd", "lib", "exp", "*.so"))):
        libs[os.path.basename(p)[:-3]] = p
    if args:
        libs = {k: v for k, v in libs.items() if k in args or k == "product"}
    nb = {2: 10, 3: 1, 4: 1, 5: 1}.get(cfg, 10)
    runs = [(name, path, None) for name, path in libs.items()] + [(f"product dbg={d}", libs["product"], d) for d in dbgs]
    for name, path, dbg in runs:
        env = dict(os.environ, STRKIT_AMD_LIB=path, STRKIT_AMD_NO_PIPE="1")   # (one call = one launch of every kernel)
        if dbg is not None:
            env["STRKIT_AMD_DBG"] = dbg
        print(f"# {name} ...", flush=True)
        out = subprocess.run([sys.executable, "-c", CODE, str(cfg), str(nb)], env=env, capture_output=True, text=True, cwd=ROOT)
        line = out.stdout.strip() or out.stderr.strip()[-300:]
        print(f"{name:16s} {line}", flush=True)
        if bench:
            out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras", "--no-e2e", "--steps", "40", "--warmup", "8",
                                  "--config", str(cfg)], env=env, capture_output=True, text=True, cwd=ROOT)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                print(f"{name:16s} bench {j['value'] / 1e6:.1f} M reads/s  ms/step {j['ms_per_step']:.4f}  parity {j['parity_check']}  fb/step {j['band_fallback_per_step']}"
                      f"  miss/step {j['window_miss_reads_per_step']}  window {j['config']['window']}  device_ms/step {j['device_ms_per_step']:.3f}"
                      f"  band_reads/step {j['band_reads_per_step']:.0f}  k_ms {j['roofline'].get('kernel_ms_overlapped', 0):.3f}", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"{name:16s} bench failed: {e} {out.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    main()
