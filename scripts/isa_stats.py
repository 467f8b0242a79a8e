"""Static picture of one fan-kernel instance from hipcc's -save-temps assembly.
usage: python scripts/isa_stats.py <dir-with-.s> [LT ZM SAVE]   (default 1 4 1)
Prints the size of the kernel, where its scratch (spill) instructions sit and, for the blocks of
the step loop (the strongly connected blocks that hold the 12 table reads of an attempt), the
instruction mix."""
import re
import sys
from collections import Counter

d = sys.argv[1]
lt, zm, sv = (sys.argv[2:5] + ["1", "4", "1"])[:3] if len(sys.argv) > 2 else ("1", "4", "1")
s = open(d + "/pgr_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
tag = f"_Z14pgr_fan_kernelILb{lt}ELi{zm}ELi{sv}E"
a = s.index("\n" + tag)
b = s.index("s_endpgm", a)
body = s[a:b].split("\n")
blocks, cur = [], None
for l in body:
    t = l.strip()
    if not t or t.startswith((";", "//")):
        continue
    m = re.match(r"^(\.LBB\d+_\d+|_Z\S+):", t)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    if t.startswith("."):
        continue
    if cur is None:
        cur = ["entry", []]
        blocks.append(cur)
    cur[1].append(t.split(";")[0].strip())
idx = {b_[0]: i for i, b_ in enumerate(blocks)}
total = sum(len(b_[1]) for b_ in blocks)
print(f"{tag}: {total} instructions in {len(blocks)} blocks")
scr = [(b_[0], sum(1 for i in b_[1] if i.startswith("scratch_"))) for b_ in blocks]
scr = [x for x in scr if x[1]]
print("scratch instructions:", sum(x[1] for x in scr), "in blocks", scr[:30])
# successors
succ = {}
for i, (name, ins) in enumerate(blocks):
    out = set()
    for t in ins:
        m = re.match(r"s_(c?branch\S*)\s+(\.LBB\d+_\d+)", t)
        if m:
            out.add(idx[m.group(2)])
    last = ins[-1] if ins else ""
    if not last.startswith("s_branch") and i + 1 < len(blocks):
        out.add(i + 1)
    succ[i] = out


def kind(t):
    op = t.split()[0]
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "other"


# the step loop: blocks with ds_read_b128 (LDS table) or global_load_dwordx4 (HBM table) inside a cycle
need = 4
hot = [i for i, b_ in enumerate(blocks) if sum(1 for t in b_[1] if t.startswith(("ds_read_b128", "global_load_dwordx4"))) >= need]
if not hot:
    # (kernels whose look-up ends in a rarely taken block -- ZM = 5 -- have one table read pair per block:
    # start from the first such block that sits in a loop, i.e. the attempt's first stage)
    need = 2
    hot = [i for i, b_ in enumerate(blocks) if sum(1 for t in b_[1] if t.startswith(("ds_read_b128", "global_load_dwordx4"))) >= need
           and sum(1 for t in b_[1] if t.startswith("v_")) >= 60][:1]
print(f"blocks with >= {need} table reads:", [(blocks[i][0], len(blocks[i][1])) for i in hot])
for i in hot[:2]:
    # walk the fall-through / branch chain from this block until we come back to it
    seen, order, j = set(), [], i
    while j not in seen:
        seen.add(j)
        order.append(j)
        nxt = sorted(succ[j])
        # prefer the path that returns to i: take a backward edge to i if present, else fall through
        if i in succ[j] and j != i:
            break
        fall = j + 1 if (j + 1) in succ[j] else (nxt[0] if nxt else None)
        if fall is None:
            break
        j = fall
    c = Counter()
    n = 0
    for j in order:
        for t in blocks[j][1]:
            c[kind(t)] += 1
            n += 1
    lit = sum(1 for j in order for t in blocks[j][1] if t.startswith("s_mov_b32") and re.search(r"0x[0-9a-f]{6,}", t))
    print(f"loop from {blocks[i][0]}: blocks {[blocks[j][0] for j in order][:12]}... {n} instructions on the fall-through path:", dict(c), "| s_mov_b32 of 32-bit literals:", lit)
    ops = Counter(t.split()[0] for j in order for t in blocks[j][1])
    print("   top opcodes:", ops.most_common(28))
    if "--path" in sys.argv:
        # the instructions of the path that are not plain fp64 arithmetic or table reads, in order, with their block
        plain = ("v_mul_f64", "v_add_f64", "v_fma_f64", "v_fmac_f64", "ds_read_b128", "v_max_f64")
        for j in order:
            for t in blocks[j][1]:
                if not t.startswith(plain):
                    print(f"      {blocks[j][0]:12s} {t}")
