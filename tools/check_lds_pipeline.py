"""Static check of the hand-pipelined LDS fragment reads in the compiled conv3x3_mfma kernels.

conv3x3_mfma.h issues its `ds_read_b128` fragment reads through inline asm several reads ahead of their use and retires
them with counted `s_waitcnt lgkmcnt(N)`.  Between the read and the wait the compiler believes the destination VGPRs
already hold the data; if register pressure makes it copy or spill such a register (v_accvgpr_write, scratch_store,
v_mov ...) before the wait, it moves stale data.  This script compiles a .hip file to gfx950 assembly and verifies, per
kernel, that no instruction touches the destination of an in-flight asm read before a wait has retired it, and that
no asm read is in flight across a loop back-edge (forward skips are covered by the linear scan).

Second rule (round 2): MATRIX-OPERAND PROVENANCE.  The wrong sums round 1 recorded for the pipelined one-wave-per-SIMD fp32
ACCIN kernel were traced (tools/gpu_accin_probe2.py + the ISA) to a register-allocation miscompile of hipcc (ROCm 7.2), not
to the pipeline: with all 512 registers in use and weights shuttling through scratch and AGPRs, the 128-bit weight fragment
w[tap 0][k-group 0] was split into {scratch slot (x, y), a141 (z), a140 (w)}; phase 0 of the three-way unrolled walk
reassembles it correctly, phases 1 and 2 reload (x, y), copy w -- and never copy z: their first MFMAs read an accumulator
register that still holds element z of ANOTHER fragment, written by phase 0.  The in-flight rule cannot see that (no LDS
destination is touched).  What makes it visible statically: the MFMAs that consume the 2 or 4 elements of one LDS fragment
(consecutive registers of one asm read, same accumulator) take the matching elements of ONE weight fragment as their other
operand.  Inside the main loop each of those weight registers is either loop-invariant (never written in the loop), or
written earlier in the same barrier-delimited phase (reloaded / copied), or carried over from another phase (a reload the
compiler hoisted).  A fragment group that MIXES "written in this phase" with "carried over from another phase" is a
half-reassembled fragment -- exactly the miscompile above -- and is reported.

usage: check_lds_pipeline.py file.hip [more.hip ...]      (exit code 1 on a violation)
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
VMC = re.compile(r"vmcnt\((\d+)\)")
VMEM_OPS = ("buffer_load", "global_load", "scratch_load", "buffer_store", "global_store", "scratch_store", "buffer_atomic",
            "global_atomic", "flat_")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def per_file_flags(path):
    """the FLAGS_<file stem> := ... line of the Makefile next to the source, so the checked code is the shipped code"""
    mk = os.path.join(os.path.dirname(os.path.abspath(path)), "Makefile")
    stem = os.path.splitext(os.path.basename(path))[0]
    if os.path.exists(mk):
        for line in open(mk):
            m = re.match(r"FLAGS_%s\s*:?=\s*(.*)" % re.escape(stem), line)
            if m:
                return m.group(1).split()
    return []


def compile_to_asm(path):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(path))] + per_file_flags(path) + ["-o", out, path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def check_asm(text):
    """returns (kernels checked, asm reads seen, list of violations)"""
    violations, kernels, nreads = [], 0, 0
    kernel, queue, in_asm, labels = None, [], False, set()   # queue: issue-ordered LDS ops, entries = (is_asm_read, dest regs, line no)
    vqueue = []   # issue-ordered vector-memory ops (vmcnt retires them in order): (is_asm_load, dest regs, line no)
    for ln, raw in enumerate(text.split("\n"), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.startswith("_Z") and raw.rstrip().split(";")[0].rstrip().endswith(":"):
            kernel, queue, vqueue, labels = raw.split(":")[0], [], [], set()
            kernels += 1
            continue
        if kernel is None or not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith("s_endpgm"):
            kernel = None
            continue
        inflight = set().union(*[q[1] for q in queue if q[0]]) if queue else set()
        vinflight = set().union(*[q[1] for q in vqueue if q[0]]) if vqueue else set()
        if line.endswith(":"):                      # label: a forward skip lands here, the linear scan covers both paths
            labels.add(line[:-1])
            continue
        if line.startswith("s_cbranch") or line.startswith("s_branch"):
            if inflight and line.split()[-1] in labels:   # backward branch (loop): nothing may be in flight
                violations.append((kernel, ln, "asm LDS read in flight across a loop back-edge: " + line))
            if vinflight and line.split()[-1] in labels:
                violations.append((kernel, ln, "asm buffer load in flight across a loop back-edge: " + line))
            continue
        m = LGKM.search(line)
        if line.startswith("s_waitcnt"):
            mv = VMC.search(line)
            if mv:
                n = int(mv.group(1))
                vqueue = vqueue[len(vqueue) - n:] if n else []
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                if n == 0:
                    queue = []
            continue
        if line.startswith("s_barrier"):
            continue
        touched = regs_of(line)
        if in_asm and line.startswith("buffer_load"):     # asm global -> register load (wgrad_mfma.hip): same contract, on vmcnt
            dest = regs_of(line.split(",")[0])
            if dest & (vinflight | inflight):
                violations.append((kernel, ln, "asm buffer load overwrites an in-flight destination: " + line))
            if (regs_of(",".join(line.split(",")[1:])) & (vinflight | inflight)):
                violations.append((kernel, ln, "asm buffer load reads an in-flight destination: " + line))
            vqueue.append((True, dest, ln))
            nreads += 1
            continue
        if touched & vinflight:
            violations.append((kernel, ln, "touches an in-flight buffer-load destination: " + line))
        if line.startswith(VMEM_OPS):
            vqueue.append((False, set(), ln))
        if in_asm and line.startswith("ds_read"):
            dest = regs_of(line.split(",")[0])
            if dest & inflight:
                violations.append((kernel, ln, "asm read overwrites an in-flight destination: " + line))
            queue.append((True, dest, ln))
            nreads += 1
            continue
        if touched & inflight:
            violations.append((kernel, ln, "touches an in-flight LDS destination: " + line))
        if line.startswith("ds_"):
            queue.append((False, set(), ln))
    return kernels, nreads, violations


AREG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


def _regs(tok):
    out = set()
    for m in AREG.finditer(tok):
        if m.group(4) is not None:
            out.add((m.group(4), int(m.group(5))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_operand_provenance(text):
    """returns (kernels with a main loop, MFMAs checked, violations): see the module docstring, second rule."""
    violations, nk, nm = [], 0, 0
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().split(";")[0].rstrip().endswith(":")]
    for st in starts:
        try:
            en = next(i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        except StopIteration:
            continue
        body = [l.split(";")[0].strip() if not l.strip().startswith(";;#") else "" for l in lines[st:en]]
        label_at = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":") and not l.startswith("_Z")}
        # outermost loop that contains MFMAs: widest backward branch span
        loop = None
        for i, l in enumerate(body):
            if l.startswith(("s_cbranch", "s_branch")):
                tgt = l.split()[-1]
                j = label_at.get(tgt)
                if j is not None and j < i and any(b.startswith("v_mfma") for b in body[j:i]):
                    if loop is None or (i - j) > (loop[1] - loop[0]):
                        loop = (j, i)
        if loop is None:
            continue
        nk += 1
        lo, hi = loop
        bars = [i for i in range(lo, hi + 1) if body[i].startswith("s_barrier")]
        bounds = [lo] + bars + [hi + 1]
        seg_of = {}
        for k in range(len(bounds) - 1):
            for i in range(bounds[k], bounds[k + 1]):
                seg_of[i] = k
        written = {}                       # reg -> set of segments where it is written inside the loop
        writes_before = {}                 # (segment, reg) -> first line of a write
        for i in range(lo, hi + 1):
            l = body[i]
            if not l or l.endswith(":") or l.startswith(("s_", ";", ".")):
                continue
            toks = l.split(None, 1)
            if len(toks) < 2:
                continue
            if toks[0].startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "global_atomic")):
                continue
            for r in _regs(toks[1].split(",")[0]):
                written.setdefault(r, set()).add(seg_of[i])
                writes_before.setdefault((seg_of[i], r), i)
        # asm-read destination tuples (fragments) by register
        frag_of = {}
        for i in range(lo, hi + 1):
            if body[i].startswith("ds_read_b128") or body[i].startswith("ds_read_b64"):
                dst = sorted(_regs(body[i].split(None, 1)[1].split(",")[0]))
                for r in dst:
                    frag_of[r] = dst[0]
        groups = {}                        # (segment, accumulator, fragment base, ordinal of that fragment's reuse) -> statuses
        seen = {}
        for i in range(lo, hi + 1):
            l = body[i]
            if l.startswith("ds_read"):
                for r in _regs(l.split(None, 1)[1].split(",")[0]):
                    seen[frag_of.get(r, r)] = seen.get(frag_of.get(r, r), 0) + 1
                continue
            if not l.startswith("v_mfma"):
                continue
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            nm += 1
            ra, rb = sorted(_regs(ops[1])), sorted(_regs(ops[2]))
            # which operand is the LDS fragment?  (weights are the A operand in conv3x3_mfma / conv_split, B in wgrad)
            fa, fb = [r for r in ra if r in frag_of], [r for r in rb if r in frag_of]
            frag, other = (fb, ra) if fb else (fa, rb)
            if not frag:
                continue
            key = (seg_of[i], ops[0], frag_of[frag[0]], seen.get(frag_of[frag[0]], 0))
            for r in other:
                segs = written.get(r)
                if not segs:
                    st_ = "invariant"
                else:
                    w = writes_before.get((seg_of[i], r))
                    st_ = "phase" if (w is not None and w < i) else "carried"
                groups.setdefault(key, []).append((st_, r, i, l))
        for key, items in groups.items():
            kinds = {k for k, _, _, _ in items}
            if "phase" in kinds and "carried" in kinds:
                for k, r, i, l in items:
                    if k == "carried":
                        violations.append((lines[st].split(":")[0], st + i + 1,
                                           f"weight register {r[0]}{r[1]} is carried over from another loop phase while the rest of its "
                                           f"fragment was rewritten in this phase: {l}"))
    return nk, nm, violations


def main(paths):
    bad = 0
    for p in paths:
        asm = compile_to_asm(p)
        kernels, nreads, violations = check_asm(asm)
        nk, nm, v2 = check_operand_provenance(asm)
        violations = violations + v2
        print(f"{os.path.basename(p)}: {kernels} kernels, {nreads} pipelined LDS reads, {nm} loop MFMAs in {nk} kernels, "
              f"{len(violations)} violations")
        for k, ln, msg in violations[:20]:
            print(f"  {k[:90]} line {ln}: {msg}")
        bad += len(violations)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
