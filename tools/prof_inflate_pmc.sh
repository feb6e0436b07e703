#!/bin/bash
# PMC counters of the device inflater (k_bgzf_inflate) on the bench's end-to-end file: separate passes, never combined with a
# trace.  usage: tools/prof_inflate_pmc.sh <tag>  -> gpurun_out/pmc_inflate_<tag>/   (from the repo root on the GPU box)
TAG=${1:-k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_inflate_$TAG
mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM"; do
  [ -n "$ONLY" ] && [[ "$set" != *"$ONLY"* ]] && continue
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "pass: $set" >> $OUT/progress.txt
  # (FETCH_SIZE and WRITE_SIZE do not fit in one pass on gfx950; a pass whose counters do not fit aborts and then hangs)
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o b -- python3 tools/e2e_wall_profile.py 10000 device > $OUT/$tag.log 2>&1 || echo "pmc $set failed" >> $OUT/progress.txt
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_bgzf_inflate" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:24s} launches {len(v):3d}  mean {sum(v) / len(v):.4g}")
PY
