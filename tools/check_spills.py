"""Lists every gfx950 kernel of the library that spills vector registers or uses scratch memory, from the AMDGPU metadata notes
of the built objects (aqualora_amd/csrc/aql_*.o: .hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes; a second).
A 14-VGPR spill in the K loop of the 256-row row-tile conv kernel cost 17 % (profiles/r02_ab_conv_row_prefetch.txt) and went
unnoticed for three commits; tests/test_abi.py runs this after the build.
usage: python tools/check_spills.py          (exit code 1 when any kernel spills VGPRs or has a scratch segment)"""
import glob, os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aqualora_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    """{mangled kernel name: {metadata key: int}} of one host object with an embedded gfx950 code object."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode != 0:     # a host-only object (aql_comm.o: RCCL binding, no device code of ours)
            return {}
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for block in re.split(r"\n\s+- ", notes):   # one YAML list item per kernel
        m = re.search(r"\.name:\s+(\S+)", block)
        if not m or ".vgpr_count" not in block:
            continue
        cur = out.setdefault(m[1], {})
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "agpr_count",
                    "group_segment_fixed_size"):
            k = re.search(rf"\.{key}:\s+(\d+)", block)
            if k:
                cur[key] = int(k[1])
    return out


def scan(objs=None):
    """-> (number of kernels, [(object, kernel, metadata)] of the ones with VGPR spills or a scratch segment)"""
    objs = objs or sorted(glob.glob(os.path.join(CSRC, "aql_*.o")))
    n, bad = 0, []
    for o in objs:
        ks = kernels_of(o)
        n += len(ks)
        bad += [(os.path.basename(o), k, d) for k, d in ks.items()
                if d.get("vgpr_spill_count", 0) or d.get("private_segment_fixed_size", 0)]
    return n, bad


def main():
    n, bad = scan()
    for o, k, d in bad:
        dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"SPILL {o}: {dn[:180]}  {d}")
    print(f"{n} kernels scanned, {len(bad)} with VGPR spills or scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
