"""Static check of the hand-pipelined LDS fragment reads in the compiled conv3x3_mfma kernels.

conv3x3_mfma.h issues its `ds_read_b128` fragment reads through inline asm several reads ahead of their use and retires
them with counted `s_waitcnt lgkmcnt(N)`.  Between the read and the wait the compiler believes the destination VGPRs
already hold the data; if register pressure makes it copy or spill such a register (v_accvgpr_write, scratch_store,
v_mov ...) before the wait, it moves stale data.  This script compiles a .hip file to gfx950 assembly and verifies, per
kernel, that no instruction touches the destination of an in-flight asm read before a wait has retired it, and that
no asm read is in flight across a loop back-edge (forward skips are covered by the linear scan).

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


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_to_asm(path):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(path)), "-o", out, path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def check_asm(text):
    """returns (kernels checked, asm reads seen, list of violations)"""
    violations, kernels, nreads = [], 0, 0
    kernel, queue, in_asm, labels = None, [], False, set()   # queue: issue-ordered LDS ops, entries = (is_asm_read, dest regs, line no)
    for ln, raw in enumerate(text.split("\n"), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.startswith("_Z") and raw.rstrip().split(";")[0].rstrip().endswith(":"):
            kernel, queue, labels = raw.split(":")[0], [], set()
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
        if line.endswith(":"):                      # label: a forward skip lands here, the linear scan covers both paths
            labels.add(line[:-1])
            continue
        if line.startswith("s_cbranch") or line.startswith("s_branch"):
            if inflight and line.split()[-1] in labels:   # backward branch (loop): nothing may be in flight
                violations.append((kernel, ln, "asm LDS read in flight across a loop back-edge: " + line))
            continue
        m = LGKM.search(line)
        if line.startswith("s_waitcnt"):
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                if n == 0:
                    queue = []
            continue
        if line.startswith("s_barrier"):
            continue
        touched = regs_of(line)
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


def main(paths):
    bad = 0
    for p in paths:
        kernels, nreads, violations = check_asm(compile_to_asm(p))
        print(f"{os.path.basename(p)}: {kernels} kernels, {nreads} pipelined LDS reads, {len(violations)} violations")
        for k, ln, msg in violations[:20]:
            print(f"  {k[:90]} line {ln}: {msg}")
        bad += len(violations)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
