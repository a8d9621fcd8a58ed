#!/bin/bash
# round 3, session 19: the overlap-save kernel's variants side by side: time, VGPRs / LDS / residency (rocprofv3's dispatch
# record), SQ_WAIT_INST_LDS and friends; and the nontemporal-store build of the default variant (libmi355dsp_nts.so: -DMDSP_IO_AUX_STORE=2)
REPO="$(pwd)"; O="$REPO/gpurun_out/ols_variants"; mkdir -p "$O"
export TUNE_OLS=0,14,16,18,19,23,29,30,33,34,36 TUNE_WELCH= TUNE_WGS= TUNE_RUNS= TUNE_ROUNDS=5
echo "== variants, default build"
python tools/tune.py 2>&1 | grep "^ols\|^copy" | tee $O/variants.log; cp gpurun_out/tune.json $O/variants.json
echo "== default variant, nontemporal stores"
for tag in "" nts "" nts; do
  TUNE_OLS=0 MDSP_LIB_TAG=$tag python tools/tune.py 2>&1 | grep "^ols fused" | sed "s/^/[${tag:-default}] /" | tee -a $O/nts.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p1 $O/p2
TUNE_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES \
    -d $O/p1 -o s -- python $REPO/tools/tune.py > $O/p1.log 2>&1
TUNE_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    -d $O/p2 -o s -- python $REPO/tools/tune.py > $O/p2.log 2>&1
cd $REPO
for p in p1 p2; do
  db=$(find $O/$p -name '*.db' | head -1)
  [ -n "$db" ] && python tools/prof_summary.py --pmc "$db" > $O/$p.json
done
rm -rf $O/p1 $O/p2
python - <<'PY'
import json
o = "gpurun_out/ols_variants/"
a, b = json.load(open(o + "p1.json")), json.load(open(o + "p2.json"))
for k in sorted(a):
    if "ols_fused_kernel" not in k: continue
    e = a[k]; c = dict(e["counters"]); c.update(b.get(k, {}).get("counters", {}))
    g = lambda n: c.get(n, float("nan"))
    print(k[17:80], "us", round(e["avg_ns"] / 1e3, 1), "vgpr", e["vgpr"], "lds", e["lds"], "wg", e["wg"],
          "| wait_lds/wave %.3f wait_any/wave %.3f valu_active/wave %.3f lds_conflict/lds_active %.3f" % (
              g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
              g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
PY
