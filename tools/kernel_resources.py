"""Prints VGPR / SGPR / LDS / scratch / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage, the flags of
emoportraits_amd/build.py).

    python tools/kernel_resources.py <file.hip> [...]      # table per file
    python tools/kernel_resources.py --audit               # every conv_inst_*.hip: the kernels hide their global loads in inline
                                                           # asm, and a register the compiler spills or copies while such a load is
                                                           # in flight is silent corruption (cdna_hip_programming.md section 5.7).
                                                           # Invariant checked in the generated ISA: NO scratch access between the
                                                           # first pinned load (prologue) and the last MFMA of a kernel.  Exit code 1 on a violation.
"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emoportraits_amd import build as B  # noqa: E402


def resources(src):
    cmd = [B.HIPCC] + B.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", name))}
            rows.append(cur)
            continue
        for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def show(row):
    return (f"{row['name'][:100]:100s} vgpr={row.get('VGPRs')} sgpr={row.get('TotalSGPRs')} "
            f"scratch={row.get('ScratchSize [bytes/lane]')} occ={row.get('Occupancy [waves/SIMD]')}")


def loop_scratch(src):
    """per kernel of `src`: (name, scratch instructions between the first and the last MFMA, scratch instructions in total)"""
    cmd = [B.HIPCC] + B.FLAGS + ["--cuda-device-only", "-S", src, "-o", "-"]
    asm = subprocess.run(cmd, capture_output=True, text=True).stdout.splitlines()
    out, name, lines = [], None, []
    for line in asm + ["\t.end_of_file -- Begin function"]:
        if "-- Begin function" in line:
            if name is not None:
                mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
                first_asm = next((i for i, l in enumerate(lines) if "#ASMSTART" in l), None)   # first pinned load (prologue)
                sc = [i for i, l in enumerate(lines) if "scratch_" in l and not l.lstrip().startswith(";")]
                lo = min(first_asm if first_asm is not None else 10 ** 9, mf[0] if mf else 10 ** 9)
                inside = [i for i in sc if mf and lo < i < mf[-1]]
                out.append((name, len(inside), len(sc)))
            m = re.search(r"Begin function (\S+)", line)
            name = m.group(1) if m else None
            lines = []
        else:
            lines.append(line)
    return out


if __name__ == "__main__":
    if sys.argv[1:] == ["--audit"]:
        files = sorted(glob.glob(os.path.join(B.CSRC, "conv_inst_*.hip")))
        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            res = list(ex.map(loop_scratch, files))
        bad = total = spilling = 0
        for f, rows in zip(files, res):
            for name, inside, anywhere in rows:
                total += 1
                spilling += anywhere > 0
                if inside:
                    bad += 1
                    nm = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
                    print(f"VIOLATION {os.path.basename(f)}: {nm[:110]}: {inside} scratch accesses inside the K loop")
        print(f"{total} kernels, {spilling} with scratch accesses outside the K loop (prologue / epilogue only), {bad} violations")
        sys.exit(1 if bad else 0)
    for src in sys.argv[1:]:
        for row in resources(src):
            print(show(row))
