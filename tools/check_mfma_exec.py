"""Lists MFMA instructions that hipcc placed under an EXEC mask without a skip branch.  MFMA ignores EXEC: a block entered
with `s_and_saveexec` and NO `s_cbranch_execz` runs its MFMAs for a wavefront whose condition is false everywhere, on whatever
its (possibly uninitialised) operand registers hold.  For wave-uniform guards written on threadIdx expressions that is a silent
wrong result: lora_down_skinny_kernel accumulated NaN for every K < 256 until `wave` became a readfirstlane (round 3).
Works on the gfx950 code objects embedded in the built aqualora_amd/csrc/aql_*.o (llvm-objdump -d; a second per object);
tests/test_abi.py runs it after the build.     usage: python tools/check_mfma_exec.py   (exit code 1 on any hit)"""
import glob, os, re, subprocess, sys, tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aqualora_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
OPENERS = ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_xor_saveexec_b64", "s_or_saveexec_b64")
STOPPERS = ("s_or_b64 exec", "s_mov_b64 exec", "s_cbranch", "s_branch", "s_endpgm", "s_setpc") + OPENERS


def scan_asm(text):
    """[(function, line number, instruction)] of MFMAs reached under a saveexec mask with no execz / execnz branch in between"""
    hits, func = [], None
    lines = [re.sub(r"\s*//.*$", "", l).strip() for l in text.splitlines()]
    for i, ln in enumerate(lines):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", ln)
        if m:
            func = m.group(1)
        if not ln.startswith(OPENERS):
            continue
        for j in range(i + 1, len(lines)):
            t = lines[j]
            if not t or t.startswith(";"):
                continue
            if t.startswith("s_cbranch_execz") or t.startswith("s_cbranch_execnz"):
                break
            if t.startswith("v_mfma"):
                hits.append((func, i + 1, t))
                break
            if t.endswith(":") or t.startswith(STOPPERS):
                break
    return hits


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode != 0:
            return None       # host-only object
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def scan(objs=None):
    """-> (number of MFMA instructions seen, [(object, function, line, instruction)])"""
    objs = objs or sorted(glob.glob(os.path.join(CSRC, "aql_*.o")))
    n, bad = 0, []
    for o in objs:
        text = disassemble(o)
        if text is None:
            continue
        n += text.count("v_mfma")
        bad += [(os.path.basename(o),) + h for h in scan_asm(text)]
    return n, bad


def main():
    n, bad = scan()
    for o, fn, line, ins in bad:
        dn = subprocess.run(["c++filt", fn or "?"], capture_output=True, text=True).stdout.strip()
        print(f"EXEC-MASKED MFMA {o}:{line}: {dn[:150]}: {ins}")
    print(f"{n} MFMA instructions, {len(bad)} under an EXEC mask without a skip branch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
