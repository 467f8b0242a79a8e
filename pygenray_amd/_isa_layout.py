"""Build step: lay the device code out so that no 8-byte instruction straddles a 32-byte fetch window.

Why.  A lone wave (the steepest rays of a fan: the fan's run time is their latency)
pays ~3.6 extra cycles for every 8-byte (VOP3 / DS / SOP-with-literal) instruction that crosses a
32-byte boundary of the instruction stream (scripts/probes/align_probe.hip: 64 x v_mul_f64 at +0:
5.2 cycles each, at +4: 6.1).  The fp64 stepper is mostly VOP3; which of its instructions straddle
depends on how many 4-byte instructions precede them, i.e. on nothing the source controls: shifting
the whole loop by ONE dword moved the lone-wave time by 3.5 % (4.88 <-> 4.71 ms).

How.  After the compiler has produced the gfx950 assembly, every 4-byte VALU instruction in `_e32`
encoding whose operands also fit the 8-byte `_e64` encoding is a free choice between 4 and 8 bytes
(same operation, same issue cost).  A dynamic programme over each kernel's linear instruction list
(state = offset mod 32) picks the encodings that minimise the number of straddling instructions;
nothing is inserted, removed or reordered.  Instruction sizes come from the assembler itself
(llvm-objdump of the unmodified object), not from rules.

This module only rewrites text; pygenray_amd/_lib.py drives the compiler around it and falls back
to the plain build if anything here raises.
"""
import re
import subprocess

WINDOW = 32

# e32 VALU mnemonics that are safe to re-encode as e64 (plain VGPR/SGPR/inline-constant operands;
# no carry-in/out forms other than v_cndmask's vcc, no SDWA/DPP, no literals)
_PROMOTABLE = re.compile(
    r"^(v_fmac_f64|v_mov_b32|v_cvt_[a-z0-9]+_[a-z0-9]+|v_rcp_f64|v_rsq_f64|v_rcp_f32|v_add_u32|v_sub_u32|"
    r"v_subrev_u32|v_cndmask_b32|v_max_i32|v_min_i32|v_max_u32|v_min_u32|v_and_b32|v_or_b32|v_xor_b32|"
    r"v_lshlrev_b32|v_lshrrev_b32|v_ashrrev_i32|v_ceil_f64|v_floor_f64|v_trunc_f64|v_fract_f64|"
    r"v_mul_f32|v_add_f32|v_exp_f32|v_log_f32|v_cmp_[a-z]+_(f64|f32|i32|u32|i64|u64))_e32$")
_INLINE_FLOATS = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0"}
_REG = re.compile(r"^(v\d+|v\[\d+:\d+\]|s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0)$")


def _operand_ok(op):
    op = op.strip()
    if _REG.match(op) or op in _INLINE_FLOATS:
        return True
    if re.match(r"^-?\d+$", op):
        return -16 <= int(op) <= 64
    return False


def promotable(text):
    parts = text.split(None, 1)
    if not _PROMOTABLE.match(parts[0]):
        return False
    if len(parts) < 2:
        return False
    ops = parts[1].split(";")[0].split(",")
    return all(_operand_ok(o) for o in ops)


def object_layout(objdump, obj):
    """{function: [(offset_in_function, size, text), ...]} from llvm-objdump -d."""
    out = subprocess.run([objdump, "-d", obj], check=True, capture_output=True, text=True).stdout
    funcs, cur, base = {}, None, 0
    for line in out.split("\n"):
        m = re.match(r"^([0-9a-f]{16}) <([^>]+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            continue
        m = re.match(r"^\t(.*?)\s*// ([0-9A-F]{12}): ((?:[0-9A-F]{8} ?)+)", line)
        if m and cur is not None:
            addr = int(m.group(2), 16)
            size = 4 * len(m.group(3).split())
            cur.append([addr - base, size, m.group(1).strip()])
    return funcs


def _is_instruction(line):
    s = line.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
        return False
    if re.match(r"^[A-Za-z_.$][\w.$]*:", s):
        return False
    return line.startswith("\t") or line.startswith(" ")


def plan_function(items):
    """items: list of ('i', size, promotable) / ('a', log2_alignment).  Returns (set of indices to
    promote, straddles before, straddles after)."""
    INF = 1 << 60
    nstate = WINDOW // 4
    # forward DP with back-pointers; cost = (straddles, promotions)
    cost = [(INF, INF)] * nstate
    cost[0] = (0, 0)
    back = []
    for kind, a, b in items:
        new = [(INF, INF)] * nstate
        bp = [None] * nstate
        for st in range(nstate):
            c = cost[st]
            if c[0] >= INF:
                continue
            off = st * 4
            if kind == "a":
                al = 1 << a
                noff = off if al <= 4 else ((off + al - 1) // al * al) % WINDOW if al < WINDOW else 0
                ns = noff // 4
                if c < new[ns]:
                    new[ns] = c; bp[ns] = (st, 0)
                continue
            for size, promo in ((a, 0),) + (((8, 1),) if (b and a == 4) else ()):
                strad = 1 if (size == 8 and off % WINDOW == WINDOW - 4) else 0
                ns = ((off + size) % WINDOW) // 4
                nc = (c[0] + strad, c[1] + promo)
                if nc < new[ns]:
                    new[ns] = nc; bp[ns] = (st, promo)
        cost = new
        back.append(bp)
    end = min(range(nstate), key=lambda s: cost[s])
    after = cost[end][0]
    promote, st = set(), end
    for k in range(len(items) - 1, -1, -1):
        pst, promo = back[k][st]
        if promo:
            promote.add(k)
        st = pst
    # straddles of the untouched layout
    off, before = 0, 0
    for kind, a, b in items:
        if kind == "a":
            al = 1 << a
            if al > 4:
                off = (off + al - 1) // al * al
            continue
        if a == 8 and off % WINDOW == WINDOW - 4:
            before += 1
        off += a
    return promote, before, after


def relayout(asm_text, layout, only=None):
    """Rewrite the assembly; returns (new_text, report) with report[function] = (before, after, promoted)."""
    lines = asm_text.split("\n")
    report = {}
    k = 0
    while k < len(lines):
        m = re.match(r"^([A-Za-z_$][\w$]*):", lines[k])
        if not (m and m.group(1) in layout and (only is None or only(m.group(1)))):
            k += 1
            continue
        name = m.group(1)
        insts = layout[name]
        j = k + 1
        items, where = [], []   # where[i] = line index of item i (instructions only)
        pos = 0                 # index into insts
        pc_relative = False     # between s_getpc_b64 and the s_addc_u32 that completes the address:
                                # the literal offsets count bytes from the s_getpc, sizes must not change
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            ln = lines[j]
            s = ln.strip()
            ma = re.match(r"^\.p2align\s+(\d+)", s)
            if ma:
                # the assembler padded with s_nop: skip them in the object's list
                al = 1 << int(ma.group(1))
                while pos < len(insts) and insts[pos][0] % al != 0 and insts[pos][2].startswith("s_nop"):
                    pos += 1
                items.append(("a", int(ma.group(1)), 0)); where.append(None)
            elif s.startswith(".rept") or s.startswith(".endr"):
                raise RuntimeError(f"{name}: .rept in device assembly is not supported")
            elif _is_instruction(ln):
                if pos >= len(insts):
                    raise RuntimeError(f"{name}: more instructions in the assembly than in the object")
                off, size, text = insts[pos]
                mn_s, mn_o = s.split()[0], text.split()[0]
                if re.sub(r"_e(32|64)$", "", mn_s) != re.sub(r"_e(32|64)$", "", mn_o):
                    raise RuntimeError(f"{name}: line {j}: '{mn_s}' does not match object '{mn_o}'")
                if mn_s == "s_getpc_b64":
                    pc_relative = True
                items.append(("i", size, size == 4 and not pc_relative and promotable(s))); where.append(j)
                if mn_s == "s_addc_u32":
                    pc_relative = False
                pos += 1
            j += 1
        if pos != len(insts):
            # trailing padding after s_endpgm is fine, anything else is not
            rest = [t for _, _, t in insts[pos:] if not t.startswith("s_nop") and not t.startswith("s_code_end")]
            if rest:
                raise RuntimeError(f"{name}: {len(rest)} object instructions not found in the assembly")
        promote, before, after = plan_function(items)
        for i in promote:
            li = where[i]
            lines[li] = re.sub(r"^(\s*\S+?)_e32\b", r"\1_e64", lines[li], count=1)
        report[name] = (before, after, len(promote))
        k = j
    return "\n".join(lines), report


def _canon(text):
    t = re.sub(r"_e(32|64)\b", "", text)
    t = re.sub(r"\s+", " ", t).strip()
    # branch displacements change with the layout; their targets are checked by the assembler
    return re.sub(r"^(s_cbranch\w*|s_branch|s_call_b64.*?,) .*", r"\1", t)


def check_same_program(before, after):
    """The re-encoded object must hold, function by function, the same instructions in the same
    order (padding s_nop aside): raises if not."""
    if set(before) != set(after):
        raise RuntimeError("functions differ after the layout pass")
    for f in before:
        a = [_canon(t) for _, _, t in before[f] if not t.startswith("s_nop")]
        b = [_canon(t) for _, _, t in after[f] if not t.startswith("s_nop")]
        if a != b:
            k = next((i for i, (p, q) in enumerate(zip(a, b)) if p != q), min(len(a), len(b)))
            raise RuntimeError(f"{f}: instruction {k} differs after the layout pass")
