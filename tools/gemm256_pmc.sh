#!/bin/bash
# (GPU) PMC passes over tools/gemm256_probe.hip: L2 hit / miss counts, HBM / fabric fetch bytes, LDS bank conflicts, MFMA-busy cycles.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/gemm256_probe.hip -o gpurun_out/gemm256_probe || exit 1
OUT=gpurun_out/g256pmc
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*" $OUT/counters.txt | sort -u | tr '\n' ' ' | fold -w 200 > $OUT/counter_names.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d $OUT/$tag -o p --output-format csv -- ./gpurun_out/gemm256_probe p > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get("Kernel_Name", "")[:40], r.get("Grid_Size", ""), r["Counter_Name"])
    a = agg.setdefault(k, [0.0, 0])
    a[0] += float(r["Counter_Value"]); a[1] += 1
for (kn, gs, cn), (v, n) in agg.items():
    if "gemm256" in kn: print(f"{kn[11:40]:30s} {cn:32s} per launch {v / n:16.1f}  ({n} launches)")
PY
  rm -rf $OUT/$tag
done
rm -f gpurun_out/gemm256_probe
