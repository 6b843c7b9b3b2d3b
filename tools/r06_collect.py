"""Copies the summaries tools/r06_refresh.sh left in gpurun_out/r06/ into profiles/r06_* and rebuilds the JSON files bench.py reads
(r06_pmc_traffic.json, r06_pmc_mfma_util.json, r06_meta.json).  Run in the build container after the gpurun call."""
import hashlib
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
for src, dst in (("insitu_r06final_summary.txt", "r06_kernel_trace_graph_step.txt"), ("insitu_r06final_shapes.txt", "r06_kernel_trace_by_shape.txt"),
                 ("insitu_r06final_sequence.txt", "r06_step_sequence.txt"), ("insitu_r06final_c3_summary.txt", "r06_kernel_trace_config3.txt"),
                 ("insitu_r06final_families.json", "r06_families_config2.json"), ("insitu_r06final_c3_families.json", "r06_families_config3.json"),
                 ("robft_r06_stats.txt", "r06_robft_stats.txt"), ("extract_b1_stats.txt", "r06_extract_b1_stats.txt"),
                 ("extract_b16_stats.txt", "r06_extract_b16_stats.txt"), ("prof_vae.txt", "r06_vae_stats.txt"),
                 ("infer_r06_stats.txt", "r06_infer_stats.txt"), ("pmc_mfma_util.txt", "r06_pmc_mfma_util.txt"),
                 ("pmc_robft.json", "r06_pmc_robft.json"), ("cmp_vendor.txt", "r06_cmp_vendor.txt"),
                 ("infer_r06_sequence.txt", "r06_infer_forward_sequence.txt")):
    shutil.copy(os.path.join(R, src), os.path.join(P, dst))


def rd(name):
    t = open(os.path.join(R, name)).read()
    return float(re.search(r"FETCH_SIZE\s+([\d.]+)", t).group(1)), float(re.search(r"WRITE_SIZE\s+([\d.]+)", t).group(1))


tr = json.load(open(os.path.join(P, "r06_pmc_traffic.json")))
for key, name in (("lora_geglu 320->2x1280 M=32768", "pmct_geglu.txt"), ("conv3x3 320->320 @64x64 B=8", "pmct_conv8.txt"),
                  ("chain a: to_out+res -> LN -> to_q, M=32768 (twin)", "pmct_chain.txt")):
    f, w = rd(name)
    tr[key].update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, traffic_bytes=int((2 * f + w) * 1024))
json.dump(tr, open(os.path.join(P, "r06_pmc_traffic.json"), "w"), indent=1)

rows, cur = {}, None
for line in open(os.path.join(P, "r06_pmc_mfma_util.txt")):
    m = re.match(r"== pmc_one.py (.*)", line)
    if m:
        cur = m[1].strip()
        continue
    m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s*$", line)
    if m and cur:
        rows.setdefault(cur, {})[m[1].strip()] = (float(m[4]), round(float(m[8]) / 100, 3))
mu = json.load(open(os.path.join(P, "r06_pmc_mfma_util.json")))
K = mu["kernels"]


def put(prefix, vals):
    key = next(k for k in K if k.startswith(prefix))
    K[key]["us"] = [v[0] for v in vals]
    K[key]["mfma_util"] = [v[1] for v in vals]


def one(section, start):   # the kernel of a section whose name starts with `start` (template argument lists grow over the rounds)
    return next(v for k, v in rows[section].items() if k.startswith(start))


for prefix in ("attn_fwd_kernel", "attn_dq_kernel", "attn_dkv_kernel"):
    put(prefix, [one("attn 4 4096 8", prefix + "<64, 48"), one("attn 8 4096 8", prefix + "<64, 48")])
put("conv_row_kernel", [rows["conv 8 64 320 320"]["aqlconvrow::conv_row_kernel<64, 4, false, false, 160, 64>"]])
put("lora_geglu256_kernel", [rows["geglu 32768 1280 320"]["aqlt256::lora_geglu256_kernel<false>"]])
put("lora_gemm_kernel", [rows["lora 32768 320 320"]["lora_gemm_kernel<128, 160, 64, 80, 2>"]])
put("chain_kernel", [rows["chain 32768"]["aqlchain::chain_kernel<2, false>"]])
for prefix, kern in (("qpre: attn_fwd_kernel", "attn_fwd_kernel"), ("qpre: attn_dq_kernel", "attn_dq_kernel"), ("qpre: attn_dkv_kernel", "attn_dkv_kernel")):
    key = next((k for k in K if k.startswith(prefix)), None)
    if key is None:
        key = prefix + "<64,48,...> on a pre-scaled q (aql_sdpa_*_qpre: the step's form at the 64 x 64 level), 4 / 8 samples"
        K[key] = {}
    vals = [one("attnq 4 4096 8", kern + "<64, 48"), one("attnq 8 4096 8", kern + "<64, 48")]
    K[key]["us"], K[key]["mfma_util"] = [v[0] for v in vals], [v[1] for v in vals]
f, q, kv = (K[next(k for k in K if k.startswith(p))] for p in ("qpre: attn_fwd_kernel", "qpre: attn_dq_kernel", "qpre: attn_dkv_kernel"))
num = f["mfma_util"][1] * f["us"][1] + q["mfma_util"][0] * q["us"][0] + kv["mfma_util"][0] * kv["us"][0]
mu["attention_64x64_time_weighted"] = round(num / (f["us"][1] + q["us"][0] + kv["us"][0]), 3)
# SURVEY 8(d)'s subset "attention linears + SDPA" at the 64 x 64 level, time-weighted with the ISOLATED launch times of the same PMC passes:
# the three SDPA kernels (forward on the twin batch of 8, dQ and dK/dV on 4 samples) + the row-resident chain that holds the attention
# projections of that level in the forward pass + the one-launch LoRA linear (the backward projections' kernel).  Per block and step:
# 1 forward, 1 dQ, 1 dK/dV, 3 chains (weighted with chain `a`'s time), 6 backward linears of 16384 rows (half the 32768-row probe's time).
ch, lo = (K[next(k for k in K if k.startswith(p))] for p in ("chain_kernel", "lora_gemm_kernel"))
terms = [(f["us"][1], f["mfma_util"][1]), (q["us"][0], q["mfma_util"][0]), (kv["us"][0], kv["mfma_util"][0]),
         (3 * ch["us"][0], ch["mfma_util"][0]), (6 * 0.5 * lo["us"][0], lo["mfma_util"][0])]
mu["attention_linears_plus_sdpa_time_weighted"] = round(sum(t * u for t, u in terms) / sum(t for t, _ in terms), 3)
mu["attention_linears_plus_sdpa_formula"] = ("sum(us x MfmaUtil) / sum(us) over: attn_fwd (8 samples), attn_dq, attn_dkv (4 samples), 3 x chain_kernel<2,false> "
                                             "(32768 rows), 6 x 0.5 x lora_gemm_kernel (32768-row probe, the backward runs 16384 rows) -- one transformer "
                                             "block of the 64 x 64 level; the projections are HBM-bound by shape (168 FLOP/B): see kernels[].hbm_view on the bench line")
json.dump(mu, open(os.path.join(P, "r06_pmc_mfma_util.json"), "w"), indent=1)

meta = json.load(open(os.path.join(P, "r06_meta.json")))
meta["files"] = {f: hashlib.sha256(open(os.path.join(ROOT, "aqualora_amd", "csrc", f), "rb").read()).hexdigest()[:16] for f in meta["files"]}
json.dump(meta, open(os.path.join(P, "r06_meta.json"), "w"), indent=1)
fam = json.load(open(os.path.join(P, "r06_families_config2.json")))
print("config 2:", fam["ms_per_step"], fam["kernel_busy_ms_per_step"], fam["launches_per_step"])
fam = json.load(open(os.path.join(P, "r06_families_config3.json")))
print("config 3:", fam["ms_per_step"], fam["kernel_busy_ms_per_step"], fam["launches_per_step"])
