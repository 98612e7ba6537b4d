#!/usr/bin/env python
"""Per-kernel MFMA utilisation from one rocprofv3 pass
   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE   (with --kernel-trace only).
MfmaUtil (rocprofv3's own derived metric, gfx94x formula) = MFMA busy cycles / (GUI-active cycles x SIMDs);
flops = MOPS_F32 x 512.  usage: pmc_mfma.py <dir> > profiles/rNN_mfma_util.json"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS, XCDS = 1024, 8


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in sorted(agg.items()):
        if "rocclr" in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
        gui = sum(c["GRBM_GUI_ACTIVE"]) / n / XCDS            # summed over the XCDs' GRBMs
        mops = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [0])) / max(1, len(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [0])))
        out[k] = {"launches": n, "mfma_busy_cycles": busy, "gui_active_cycles_per_xcd": gui,
                  "mfma_util": busy / (gui * SIMDS) if gui else None, "mfma_flops_per_launch": mops * 512.0}
    json.dump({"note": "one rocprofv3 --pmc pass (kernel-trace only); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs), "
                       "the gfx94x formula rocprofv3 falls back to on gfx950; flops = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
