"""Lists every gfx950 kernel of the library that spills registers or uses scratch (hipcc -Rpass-analysis=kernel-resource-usage).
A 14-VGPR spill in the K loop of the 256-row row-tile conv kernel cost 17 % (profiles/r02_ab_conv_row_prefetch.txt); run this
after touching a kernel.  usage: python tools/check_spills.py [file.hip ...]   (exit code 1 when any kernel spills)"""
import glob, os, re, subprocess, sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aqualora_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics".split()


def scan(path):
    out = subprocess.run(["hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull],
                         capture_output=True, text=True, cwd=CSRC).stderr
    kernels, name = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m[1]
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and name:
            kernels[name][m[1].strip()] = int(m[2])
    return kernels


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "aql_*.hip")))
    bad = 0
    for f in files:
        ks = scan(f)
        for n, d in ks.items():
            hard = d.get("VGPRs Spill", 0) or d.get("ScratchSize [bytes/lane]", 0)
            if hard or d.get("SGPRs Spill", 0):   # SGPR spills go to VGPR lanes (v_writelane): reported, not counted
                dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
                print(f"{'SPILL' if hard else 'note (SGPR -> VGPR lanes)'} {os.path.basename(f)}: {dn[:160]}  {d}")
                bad += 1 if hard else 0
        print(f"{os.path.basename(f)}: {len(ks)} kernels scanned")
    print("no kernel spills" if not bad else f"{bad} kernels spill")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
