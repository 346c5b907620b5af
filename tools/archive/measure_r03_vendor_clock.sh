#!/bin/bash
# Clock (GRBM_GUI_ACTIVE / 8 / duration) and cycles of the vendor GEMM kernel (bare bf16 product + bias) and of gemm_pp (with its fused
# epilogue) on the four batch-32 SAM-B block shapes, same box. One rocprofv3 --pmc pass per kernel and shape (+ --kernel-trace for time).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vclk; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
while read name M N K mode; do
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/lib_$name -- python3 $R/tools/lib_gemm_one.py $M $N $K > /dev/null 2>$O/lib_$name.err || { echo "FAILED lib $name"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pp_$name -- python3 $R/tools/gemm_one.py $M $N $K 0 $mode > /dev/null 2>$O/pp_$name.err || { echo "FAILED pp $name"; exit 1; }
done <<'LIST'
qkv 131072 2304 768 plain
proj 131072 768 768 res
lin1 131072 3072 768 gelu
lin2 131072 768 3072 res
LIST
cd $R
python3 - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/vclk")
for name in ("qkv", "proj", "lin1", "lin2"):
    for t in ("lib", "pp"):
        d = f"{O}/{t}_{name}"
        cyc = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
               if ("Cijk" in r["Kernel_Name"] or "gemm_pp" in r["Kernel_Name"]) and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
        dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))
                     if "Cijk" in r["Kernel_Name"] or "gemm_pp" in r["Kernel_Name"])
        if not cyc or not dur: continue
        c = sum(cyc) / len(cyc) / 8.0; us = dur[len(dur) // 2] / 1e3
        print(json.dumps(dict(shape=name, kernel="vendor (bare product + bias, bf16 C)" if t == "lib" else "gemm_pp (fused epilogue)", us=round(us, 1), kcycles=round(c / 1e3, 1), ghz=round(c / us / 1e3, 3))))
PY
