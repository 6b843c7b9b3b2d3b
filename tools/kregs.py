"""usage: python tools/kregs.py <object> [substring]: VGPR / spill / scratch metadata of the kernels of one built object."""
import subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_spills as C
for k, v in C.kernels_of(sys.argv[1]).items():
    d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if len(sys.argv) < 3 or sys.argv[2] in d:
        print(d[:100], v)
