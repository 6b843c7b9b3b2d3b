"""A/B of the rob-finetune decoder step (bench.py --mode robft, BASELINE config 5) with module attributes of aqualora_amd.decoder
flipped: python tools/ab_robft.py FUSE_BN_RES=0 FUSE_BN_RES=1 ...  (each variant twice, interleaved; ms per step)."""
import io, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, runpy
import aqualora_amd.decoder as D
for kv in {kv!r}.split(","):
    k, v = kv.split("=")
    setattr(D, k, bool(int(v)))
sys.argv = ["bench.py", "--mode", "robft", "--steps", "10", "--warmup", "3"]
runpy.run_path("bench.py", run_name="__main__")
"""
for rnd in (1, 2):
    for kv in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", code.format(kv=kv)], cwd=ROOT, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else None
        print(f"{kv} round {rnd}: {ms} ms per step", flush=True)
        if ms is None:
            print(out.stderr[-800:])
