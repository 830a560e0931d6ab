R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_tn -- python3 $R/tools/diag/wgrad_bf16_probe.py > $O/pmc_tn.log 2>&1 )
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r05/pmc_tn/*/*_counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(max(fs, key=__import__("os").path.getmtime))):
    if "linear_tn_bf16" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        acc[r["Dispatch_Id"]]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
big = [v for v in acc.values() if v["_ns"] > 900000]
for k in big[0]:
    print(k, sum(v[k] for v in big) / len(big))
PY
