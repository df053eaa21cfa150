"""Developer: static check of a gfx9-family ISA listing for VGPR reads of a vector-memory load's destination that no s_waitcnt
vmcnt() covers (straight-line analysis per basic block; loads AND stores count, returns are in order).
usage: waitcnt_check.py file.s [kernel-name-substring]"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


kernel, active, queue, nbad = None, False, [], 0
for ln, line in enumerate(text, 1):
    s = line.split(";")[0].strip()
    if not s:
        continue
    if s.endswith(":"):
        if not s.startswith(".L"):
            kernel = s[:-1]
            active = want in kernel
        if "--linear" not in sys.argv:
            queue = []        # block boundary: forget (per straight-line block); --linear: keep (layout order, approximate)
        continue
    if not active or s.startswith("."):
        continue
    op, _, rest = s.partition(" ")
    ops = [t.strip() for t in rest.split(",")]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", rest)
        if m:
            n = int(m.group(1))
            queue = queue[len(queue) - n:] if n < len(queue) else queue
        elif re.fullmatch(r"\d+|0x[0-9a-f]+", rest.strip()):
            queue = []
        continue
    is_vmem = op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load"))
    is_store = op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic"))
    pending = set().union(*[q for q in queue if q]) if queue else set()
    if is_vmem:
        srcs = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
        hit = srcs & pending
        dst = regs(ops[0])
        if hit:
            nbad += 1
            print(f"{kernel}: line {ln}: {s}\n    reads v{sorted(hit)} with a load still in flight")
        queue.append(dst)
    elif is_store:
        srcs = set().union(*[regs(t) for t in ops])
        hit = srcs & pending
        if hit:
            nbad += 1
            print(f"{kernel}: line {ln}: {s}\n    reads v{sorted(hit)} with a load still in flight")
        queue.append(set())
    else:
        rd = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
        wr = regs(ops[0]) if ops else set()
        # (MFMA / FMA style ops also read their destination when it appears among the sources: covered by ops[1:])
        hit = (rd | wr) & pending
        if hit:
            nbad += 1
            print(f"{kernel}: line {ln}: {s}\n    touches v{sorted(hit)} with a load still in flight ({len(queue)} outstanding)")
print(f"{nbad} uncovered uses")
