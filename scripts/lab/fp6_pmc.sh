#!/bin/bash
# PMC passes over the FP6 lab harness: LDS pipe, VMEM issue, waits (per-dispatch counters of mfma_filter_kernel_f6 and of the int8 v7 kernel beside it)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/fp6_pmc
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/fp6_pmc/p$i -o p$i --output-format csv -- $R/scripts/lab/fp6_filter_lab 1456128 > $R/gpurun_out/fp6_pmc/p$i.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in sorted(glob.glob(R+"/gpurun_out/fp6_pmc/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "f6" in k or "v7" in k:
            agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(k, {c: round(sum(x[-3:])/len(x[-3:])) for c,x in v.items()})
P
