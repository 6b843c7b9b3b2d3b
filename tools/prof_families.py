"""Per-family roofline table of the captured PPFT step from a rocprofv3 kernel trace (rocpd sqlite db of tools/insitu_profile.sh).

    python tools/prof_families.py <db> <nsteps> <batch> <rank>  ->  JSON on stdout (committed as profiles/r03_families_config<N>.json,
                                                                     read by bench.py into the bench line's `families`)

Kernel names are binned into families; each family's ALGORITHMIC work per step comes from SURVEY.md section 8(d) / BASELINE.md
section 2 (per image: conv3x3 400.3, feed-forward linears 153.5, SDPA 126.1, attention linears 79.7, 1x1 conv 43.6 GFLOP forward;
LoRA 0.7006*r GFLOP forward; GroupNorm >= 180 MB, LayerNorm >= 139 MB per pass), with the step = clean forward + watermarked
forward + backward-data (GEMMs: 3x forward; attention backward = 2.5x forward; norms: backward = 1.5 passes):
    linear     3 * (153.5 + 79.7 + 43.6) * B  +  2 * 0.7006 * r * B   GFLOP   (LoRA branch forward + backward-data)
    conv3x3    3 * 400.3 * B                                           GFLOP
    attention  (2 + 2.5) * 126.1 * B                                   GFLOP
    weight_grad  0.7006 * r * B                                        GFLOP   (dA, dB of the 192 sites)
    groupnorm  3.5 * 180 MB * B,  layernorm  3.5 * 139 MB * B          bytes
Peaks: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM (MI355X_MICROARCH.md)."""
import json
import re
import sqlite3
import sys

FAMILIES = [   # first match wins
    ("attention", r"attn_"),
    ("conv3x3", r"conv_row_kernel|ConvFwdLoader|ConvBwdLoader"),
    ("weight_grad", r"gemm_tn|lora_ds"),
    ("linear", r"lora_gemm_kernel|lora_geglu256_kernel|lora_down|gemm_kernel|chain_kernel|chain_wide_kernel"),   # (the chain kernels carry their LayerNorms: round 5)
    ("groupnorm", r"gn_"),
    ("layernorm", r"ln_kernel"),
    ("splitk_finalize", r"splitk_finalize"),
    ("optimizer", r"adamw|sumsq|cast_transpose"),
    ("elementwise_ours", r"cat_channels|split_channels|geglu|add_noise|mse|mapper|secret|upsample|timestep"),
    ("torch_elementwise", r"at::native|rocclr|copyBuffer"),
]


def main():
    db, nsteps, B, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, x in enumerate(rows) if "adamw_kernel" in x[0]][1::2]    # two adamw launches per step
    lo = marks[-nsteps - 1] + 1 if len(marks) > nsteps else 0
    sel = rows[lo:marks[-1] + 1]
    span = (sel[-1][2] - sel[0][1]) / 1e6 / nsteps
    agg = {}
    for n, s, e in sel:
        fam = next((f for f, pat in FAMILIES if re.search(pat, n)), "other")
        a = agg.setdefault(fam, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    busy = sum(v[1] for v in agg.values()) / nsteps
    work = {"linear": (3 * (153.5 + 79.7 + 43.6) * B + 2 * 0.7006 * r * B, "GFLOP"), "conv3x3": (3 * 400.3 * B, "GFLOP"),
            "attention": (4.5 * 126.1 * B, "GFLOP"), "weight_grad": (0.7006 * r * B, "GFLOP"),
            "groupnorm": (3.5 * 0.180 * B, "GB"), "layernorm": (3.5 * 0.139 * B, "GB")}
    out = []
    for fam, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        ms /= nsteps
        row = {"family": fam, "ms_per_step": round(ms, 4), "launches": round(cnt / nsteps, 1), "share": round(ms / busy, 4)}
        if fam in work:
            w, unit = work[fam]
            if unit == "GFLOP":
                row.update(work=round(w, 1), work_unit="GFLOP", achieved=round(w / ms, 1), achieved_unit="TFLOP/s", peak=2500.0,
                           frac=round(w / ms / 2500.0, 4), bound="mfma")
            else:
                row.update(work=round(w, 3), work_unit="GB", achieved=round(w / ms * 1e3, 1), achieved_unit="GB/s", peak=8000.0,
                           frac=round(w / ms * 1e3 / 8000.0, 4), bound="hbm")
        out.append(row)
    print(json.dumps({"_how": "tools/prof_families.py over the rocprofv3 --kernel-trace of `python bench.py` (captured step); "
                              "algorithmic work per family from SURVEY.md 8(d)", "batch": B, "rank": r, "steps_analysed": nsteps,
                      "ms_per_step": round(span, 3), "kernel_busy_ms_per_step": round(busy, 3),
                      "launches_per_step": round(sum(v[0] for v in agg.values()) / nsteps), "families": out}, indent=1))


if __name__ == "__main__":
    main()
