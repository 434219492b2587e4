"""Prints VGPR / SGPR / LDS / scratch / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage, the flags of
emoportraits_amd/build.py).

    python tools/kernel_resources.py <file.hip> [...]      # table per file
    python tools/kernel_resources.py --audit [-DNAME=V]    # every conv_inst_*.hip: the kernels hide their global loads in inline
                                                           # asm, and a register the compiler spills or copies while such a load is
                                                           # in flight is silent corruption (cdna_hip_programming.md section 5.7).
                                                           # Invariant checked in the generated ISA: NO scratch access between the
                                                           # first pinned load (prologue) and the last MFMA of a kernel, and no
                                                           # instruction touches the destination of a load that the listing's
                                                           # vmcnt waits have not yet covered; and no vector-memory
                                                           # instruction of an asm statement reads a scalar register
                                                           # within five wait states of a vector-ALU write of it (reloads
                                                           # of spilled scalars).  Exit code 1 on a violation.
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
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(src), []) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
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


_VREG = re.compile(r"\b([va])(\d+)\b")
_VRANGE = re.compile(r"\b([va])\[(\d+):(\d+)\]")


def _vregs(text):
    """vector registers named in an operand list; accumulation registers a<n> as 1000 + n (pinned loads may target them)"""
    regs = {int(n) + (1000 if k == "a" else 0) for k, n in _VREG.findall(text)}
    for k, a, b in _VRANGE.findall(text):
        regs.update(range(int(a) + (1000 if k == "a" else 0), int(b) + 1 + (1000 if k == "a" else 0)))
    return regs


def inflight_reads(lines, walk_epilogue=False):
    """Instructions that touch the destination VGPR of a global load which, by the issue order and the vmcnt waits of the
    listing, may still be in flight.  The kernels issue loads in inline asm and wait with hand-counted `s_waitcnt vmcnt(N)`:
    a register copy or a too-small N corrupts data silently, so the listing itself is checked.  The K loop is walked
    twice, the second time with the state the first pass left behind: loads stay in flight across the back edge.  Only
    loads issued from inline asm are tracked (the compiler waits for its own), and only between the first asm statement and the last MFMA: prologue and K loop, where the control
    flow is a single path plus exec-masked skips (the epilogue is compiler-scheduled, branchy code without pinned loads)."""
    in_asm, flag = [], False
    for l in lines:
        if "#ASMSTART" in l:
            flag = True
        in_asm.append(flag)
        if "#ASMEND" in l:
            flag = False
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    first = next((i for i, l in enumerate(lines) if "#ASMSTART" in l), None)
    if first is None or not mf:
        return []
    lo, hi = first, mf[-1]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    # outstanding VMEM operations in issue order: [sequence number, destination registers, definite]; `definite` is False
    # for an operation inside a conditionally skipped region (exec-masked piece, wave-uniform branch): some waves do not
    # issue it, so it must not be counted as "newer" when a wait is evaluated for an older load
    state = {"seq": 0, "ops": [], "skip": (0, 0)}
    hits = []

    def step(i):
        t = lines[i].split(";")[0].strip()
        if not t or t.startswith((".", "#")) or t.endswith(":"):
            return
        op = t.split()[0]
        lo_, hi_ = state["skip"]
        definite = not lo_ < i < hi_
        if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in t):
            state["ops"].append([state["seq"], set(), definite])
            state["seq"] += 1
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            state["ops"].append([state["seq"], _vregs(t.split()[1].rstrip(",")) if in_asm[i] else set(), definite])
            state["seq"] += 1
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
            state["ops"].append([state["seq"], set(), definite])
            state["seq"] += 1
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                # vmcnt(n): all but the newest n operations have completed.  An operation is certainly complete when at
                # least n DEFINITE operations are newer than it.
                ops = state["ops"]
                state["ops"] = [o for k, o in enumerate(ops) if sum(1 for o2 in ops[k + 1:] if o2[2]) < n] if n else []
        else:
            busy = set().union(*[o[1] for o in state["ops"]]) if state["ops"] else set()
            used = _vregs(t) & busy
            if used:
                hits.append((i + 1, t, sorted(used)))

    # walk the control flow: unconditional branches are followed, backward conditional branches are taken until every
    # line of the loop has been visited twice, forward conditional branches (exec-masked skips, loop exits) fall through
    visits, pc = {}, lo
    while lo <= pc < len(lines) and pc <= hi + 40 and visits.get(pc, 0) < 2:
        visits[pc] = visits.get(pc, 0) + 1
        step(pc)
        t = lines[pc].split(";")[0].strip()
        m = re.match(r"s_(c?branch)\w*\s+(\.LBB\d+_\d+)", t)
        tgt = labels.get(m.group(2)) if m else None
        if tgt is not None and m.group(1) == "cbranch" and pc < tgt <= hi:   # (a target past the last MFMA is the loop exit)
            state["skip"] = (pc, tgt)
        if tgt is not None and (m.group(1) == "branch" or tgt < pc) and lo <= tgt and visits.get(tgt, 0) < 2:
            pc = tgt
        elif t.startswith("s_endpgm"):
            break
        else:
            pc += 1
    # Behind the last MFMA: a persistent block issues the pinned loads of its NEXT item in front of the epilogue
    # (conv_igemm_bf16x3.h, EMO_S_PREFETCH_NEXT); they land at the top of the next iteration (vmcnt(0)).  The epilogue is branchy
    # compiler-scheduled code: every path from such a load is walked (both outcomes of a conditional branch) until the in-flight
    # set is empty, with the same rule -- nothing may touch a destination that the vmcnt waits on the path have not covered.
    import copy
    starts = [i for i in range(hi + 1, len(lines)) if in_asm[i] and lines[i].split(";")[0].strip().startswith(("buffer_load", "global_load"))
              and not lines[i].split(";")[0].strip().startswith("global_load_lds") and " lds" not in lines[i]]
    seen = set()
    work = []
    if starts and walk_epilogue:
        # start at the first pinned load of every straight run of them (a run = the asm loads of one prefetch site)
        runs = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 400]
        for r in runs:
            work.append((r, {"seq": 0, "ops": [], "skip": (0, 0)}))
    steps = 0
    while work and steps < 400000:
        pc, st = work.pop()
        state = st
        while 0 <= pc < len(lines) and steps < 400000:
            steps += 1
            key = (pc, frozenset().union(*[frozenset(o[1]) for o in state["ops"]]) if state["ops"] else frozenset())
            if key in seen:
                break
            seen.add(key)
            step(pc)
            t = lines[pc].split(";")[0].strip()
            if t.startswith("s_endpgm"):
                break
            if pc not in starts and not state["ops"] and pc > starts[0] + 1 and not any(pc < s_ <= pc + 400 for s_ in starts):
                break                                    # everything has landed and no further pinned load is near
            m = re.match(r"s_(c?branch)\w*\s+(\.LBB\d+_\d+)", t)
            tgt = labels.get(m.group(2)) if m else None
            if tgt is not None and m.group(1) == "branch":
                pc = tgt
            elif tgt is not None:
                work.append((tgt, copy.deepcopy(state)))
                pc += 1
            else:
                pc += 1
    return hits


def sgpr_hazards(lines):
    """Vector-memory instructions inside asm statements that read a scalar register fewer than five wait states behind a
    vector-ALU write of it (v_readlane reloads of spilled scalars, v_readfirstlane, v_cmp): the compiler pads its own
    instructions, not the contents of an asm statement.  Straight-line scan (a branch target is assumed to arrive with the
    window the fall-through path leaves: conservative in the common case of a reload right in front of the statement)."""
    def sregs(t):
        r = {int(x) for x in re.findall(r"\bs(\d+)\b", t)}
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", t):
            r.update(range(int(a), int(b) + 1))
        return r
    hits, recent, in_asm = [], [], False
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ws = int(t.split()[1]) + 1 if op == "s_nop" else 1
        if in_asm and re.match(r"(buffer|global|flat|scratch)_(load|store|atomic)", op):
            used = sregs(t)
            for age, regs in recent:
                if used & regs and age < 5:
                    hits.append((i + 1, t, sorted(used & regs), age))
        recent = [(a + ws, r) for a, r in recent if a + ws < 8]
        if op.startswith("v_"):
            # ANY vector-ALU instruction with a scalar destination is a writer: lane reads and compares (first operand), and the
            # carry / scale outputs of v_add_co / v_sub_co / v_addc_co / v_mad_u64 / v_div_scale (second operand; vcc has no number
            # and is never the scalar operand of a pinned load)
            ops_ = [o.strip() for o in t[len(op):].split(",")]
            if re.match(r"v_readlane_b32|v_readfirstlane_b32|v_cmp", op):
                w = sregs(ops_[0]) if ops_ else set()
            elif re.search(r"_co_|v_mad_[ui]64_[ui]32|v_div_scale", op):
                w = sregs(ops_[1]) if len(ops_) > 1 else set()
            else:
                w = set()
            if w:
                recent.append((0, w))
    return hits


def loop_scratch(src):
    """per kernel of `src`: (name, scratch instructions between the first pinned load and the last MFMA, scratch instructions in
    total, reads of possibly-in-flight load destinations)"""
    # (the per-file flags of the build: the listing audited is the code that ships)
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(src), []) + ["--cuda-device-only", "-S", src, "-o", "-"]
    asm = subprocess.run(cmd, capture_output=True, text=True).stdout.splitlines()
    out, name, lines = [], None, []
    for line in asm + ["\t.end_of_file -- Begin function"]:
        if "-- Begin function" in line:
            if name is not None:
                mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
                first_asm = next((i for i, l in enumerate(lines) if "#ASMSTART" in l), None)   # first pinned load (prologue)
                sc = [i for i, l in enumerate(lines) if "scratch_" in l and not l.lstrip().startswith(";")]
                # scratch traffic of the K loop proper (first to last MFMA).  Spills in the prologue are judged by
                # inflight_reads alone: it walks the prologue too, counts the compiler's own scratch operations in the
                # vmcnt model, and flags a copy or spill of any register whose pinned load may not have landed
                lo = mf[0] if mf else 10 ** 9
                inside = [i for i in sc if mf and lo < i < mf[-1]]
                out.append((name, len(inside), len(sc), inflight_reads(lines, walk_epilogue="conv_igemm_bf16x3" in name)
                            + [(ln, t + "   [scalar operand %s written by the vector ALU %d wait states before]" % (r, a), [])
                               for ln, t, r, a in sgpr_hazards(lines)]))
            m = re.search(r"Begin function (\S+)", line)
            name = m.group(1) if m else None
            lines = []
        else:
            lines.append(line)
    return out


if __name__ == "__main__":
    if sys.argv[1:2] == ["--audit"] and all(a.startswith("-D") for a in sys.argv[2:]):
        B.FLAGS = B.FLAGS + sys.argv[2:]          # (A/B builds: python -m emoportraits_amd.build --variant x NAME=V  <->  --audit -DNAME=V)
        # (the sampler's tile kernels fill LDS with global_load_lds off a scalar base as well: same hazard class, same scan;
        # they have no MFMA, so only the scalar-operand rule applies to them)
        files = sorted(glob.glob(os.path.join(B.CSRC, "conv_inst_*.hip")) + glob.glob(os.path.join(B.CSRC, "gs3d_tile_pad_*.hip"))
                       + [os.path.join(B.CSRC, "grid_sample3d.hip")])
        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            res = list(ex.map(loop_scratch, files))
        bad = total = spilling = 0
        for f, rows in zip(files, res):
            for name, inside, anywhere, racy in rows:
                total += 1
                spilling += anywhere > 0
                if inside or racy:
                    bad += 1
                    nm = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
                    if inside:
                        print(f"VIOLATION {os.path.basename(f)}: {nm[:110]}: {inside} scratch accesses inside the K loop")
                    for ln, t, regs in racy[:4]:
                        what = f"touches v{regs} while its load may be in flight" if regs else "reads a scalar register too early"
                        print(f"VIOLATION {os.path.basename(f)}: {nm[:110]}: '{t}' {what}")
        print(f"{total} kernels, {spilling} with scratch accesses outside the K loop (prologue / epilogue only), {bad} violations")
        sys.exit(1 if bad else 0)
    for src in sys.argv[1:]:
        for row in resources(src):
            print(show(row))
