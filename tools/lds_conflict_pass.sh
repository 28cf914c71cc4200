#!/bin/bash
# (GPU) one PMC pass of the bench's edit: LDS bank-conflict cycles beside all LDS-array cycles, per kernel (SQ_LDS_BANK_CONFLICT /
# SQ_LDS_IDX_ACTIVE; MI355X_MICROARCH "LDS": ds_read_b128 is served in four 16-lane groups over 64 banks).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/ldsconf
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/p -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-configs --denoise-steps 2 > /dev/null 2> $OUT/log.txt
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
    kn = re.sub(r"\(.*", "", kn)[:70]
    a = agg.setdefault(kn, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n"] += 0.25
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"])
print(f"{'kernel':70s} {'launches':>8s} {'LDS active Mcyc':>16s} {'conflict Mcyc':>14s} {'conflict share':>14s}")
for kn, a in rows[:25]:
    act, con = a["SQ_LDS_IDX_ACTIVE"], a["SQ_LDS_BANK_CONFLICT"]
    print(f"{kn:70s} {int(a['n']):8d} {act / 1e6:16.1f} {con / 1e6:14.1f} {con / max(act, 1):14.3f}")
PY
rm -rf $OUT/p
