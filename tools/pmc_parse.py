"""Average the per-dispatch counters of rocprofv3 --pmc csv output for the gemm kernels (skips the first dispatch)."""
import sys, glob, csv, collections, re
prefix, n = sys.argv[1], int(sys.argv[2])
for i in range(1, n + 1):
    files = glob.glob(f"{prefix}{i}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(f"pass {i}: no counter csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
        if not any(t in k for t in ("gemm_kernel", "geglu256_kernel", "conv_row_kernel", "attn_", "gemm_tn_tr", "chain_kernel", "chain_wide_kernel")): continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(f"pass {i}: {k[:90]}")
        for c, v in cs.items():
            v = v[1:] if len(v) > 1 else v
            print(f"    {c:44s} {sum(v)/len(v):16.1f}")
