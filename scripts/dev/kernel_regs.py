"""Register / LDS / scratch figures of the gfx950 kernels in libspx.so, read from the code objects' metadata
(no GPU needed):  python scripts/dev/kernel_regs.py [substring ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(so=None):
    so = so or os.path.join(ROOT, "spearmint_amd", "libspx.so")
    work = tempfile.mkdtemp(prefix="spx_regs_")
    cp = os.path.join(work, "lib.so")
    with open(so, "rb") as a, open(cp, "wb") as b:
        b.write(a.read())
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], stdout=subprocess.DEVNULL, cwd=work)
    out = {}
    for f in os.listdir(work):
        if "amdgcn" not in f:
            continue
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(work, f)]).decode()
        # one YAML list item per kernel, keys in alphabetical order: .agpr_count / .group_segment_fixed_size come BEFORE .name
        cur = None
        for raw in notes.splitlines():
            line = raw.strip()
            if line.startswith("- .") and not raw.startswith("      "):     # a new item of amdhsa.kernels (args are indented deeper)
                cur = {}
                line = line[2:]
            if cur is None:
                continue
            if line.startswith(".name:"):
                out[line.split()[-1]] = cur
            for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size",
                        ".vgpr_spill_count", ".sgpr_spill_count"):
                if line.startswith(key + ":"):
                    cur[key[1:]] = int(line.split()[-1])
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    for name, v in sorted(kernel_table().items()):
        if not pats or any(p in name for p in pats):
            print("%-90s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %d spill %d" % (
                name[:90], v.get("vgpr_count", -1), v.get("agpr_count", -1), v.get("sgpr_count", -1),
                v.get("group_segment_fixed_size", -1), v.get("private_segment_fixed_size", -1), v.get("vgpr_spill_count", 0)))
