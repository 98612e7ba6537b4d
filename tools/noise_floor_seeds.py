"""The accuracy statement of README / DESIGN section 5 over SEVERAL problems instead of bench.py's one: for each seed a fresh synthetic
matrix (4 096 cells x 20 000 genes, the metric's shapes: K = 40 sub-nets, D ~ 2 400, O = 512, batch 64) and fresh target / predictor lists,
bench.accuracy_pair's four runs -- the HIP engine with all 40 sub-nets (the timed job's kernels), its 5-sub-net share (the resident
kernel), the CPU port in float32 and in float64 (4 sub-nets) -- for the metric's 18 epochs, and the element-wise distance of each float32
evaluation from the float64 one.  The claim under test: the HIP path is no further from the float64 trajectory than the float32 CPU port is.
    python tools/noise_floor_seeds.py [seeds] [epochs] [learning rate]        (GPU box; ~30 s per seed; defaults 6, 18, 1e-4 = bench.py's)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench      # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 18
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
    cfg = bench.CONFIGS["cfg3"]
    rows = []
    print("learning rate %g, %d epochs" % (lr, epochs))
    print("seed | rms rel vs fp64: HIP(40)  HIP(5 resident)  fp32 port | outside 1e-4 vs fp64: HIP  fp32 port | after 1 epoch: HIP vs fp32 port outside | val loss rel diff vs fp64: HIP  fp32 port | sub-nets off (rms > 1e-3): HIP  fp32 port")
    for s in range(seeds):
        norm = bench.synth_counts(4096, cfg["g"], seed=100 + s)
        targets, preds = bench.synth_indices(cfg["g"], cfg["O"], seed=100 + s)
        a = bench.accuracy_pair(cfg, targets, preds, norm, epochs, lr, n_cells=4096, n_subnets=4)
        iv = a["imputed_values"]
        f = iv["log1p_space"]
        e1 = iv["by_epoch"]["1"]["hip_vs_cpu_port"]["outside_tolerance"]
        off_h = sum(1 for d in iv["rms_rel_per_subnet"] if d["hip_vs_cpu_port_fp64"] > 1e-3)
        off_c = sum(1 for d in iv["rms_rel_per_subnet"] if d["cpu_port_fp32_vs_fp64"] > 1e-3)
        r = dict(seed=100 + s, hip=f["hip_vs_cpu_port_fp64"]["rms_rel"], share=f["hip_share_vs_cpu_port_fp64"]["rms_rel"], port=f["cpu_port_fp32_vs_fp64"]["rms_rel"],
                 hip_out=f["hip_vs_cpu_port_fp64"]["outside_tolerance"], port_out=f["cpu_port_fp32_vs_fp64"]["outside_tolerance"], e1_out=e1,
                 hip_val=a["relative_difference_vs_fp64"]["hip"]["val_loss"], port_val=a["relative_difference_vs_fp64"]["cpu_port"]["val_loss"],
                 off_hip=off_h, off_port=off_c, n_sub=len(iv["rms_rel_per_subnet"]))
        rows.append(r)
        print("%4d | %.2e  %.2e  %.2e | %.3f  %.3f | %.4f | %.1e  %.1e | %d/%d  %d/%d" % (r["seed"], r["hip"], r["share"], r["port"], r["hip_out"], r["port_out"], r["e1_out"],
                                                                                   r["hip_val"], r["port_val"], off_h, r["n_sub"], off_c, r["n_sub"]))
        sys.stdout.flush()
    g = lambda k: np.array([r[k] for r in rows])
    print("median rms rel vs fp64: HIP %.2e, HIP share %.2e, fp32 port %.2e; HIP <= fp32 port in %d of %d seeds; HIP <= 2 x fp32 port in %d of %d" % (
        np.median(g("hip")), np.median(g("share")), np.median(g("port")), int(np.sum(g("hip") <= g("port"))), seeds, int(np.sum(g("hip") <= 2 * g("port") + 1e-7)), seeds))
    print("sub-nets whose final predictions are off the fp64 run by rms > 1e-3: HIP %d of %d, fp32 port %d of %d; after ONE epoch HIP vs fp32 port outside 1e-4: max %.4f" % (
        int(g("off_hip").sum()), int(g("n_sub").sum()), int(g("off_port").sum()), int(g("n_sub").sum()), float(g("e1_out").max())))
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
