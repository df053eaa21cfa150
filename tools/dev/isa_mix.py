"""Static instruction mix of every loop of one kernel in a hipcc -S listing (developer tool).
usage: isa_mix.py listing.s mangled_kernel_name"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
name = sys.argv[2]
a = s.index("\n" + name + ":")
b = s.index("s_endpgm", a)
body = s[a:b].splitlines()
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def mix(lines):
    c = Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith(("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag")):
            c["trans"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        else:
            c["other"] += 1
    return dict(c)


print("whole kernel:", len(body), mix(body))
for st, en in loops:
    print(f"loop lines {st}..{en} ({en - st}):", mix(body[st:en]))
