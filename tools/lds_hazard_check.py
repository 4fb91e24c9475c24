"""Developer tool: scan a gfx950 assembly listing (hipcc -S --cuda-device-only) for instructions that touch a register
while an LDS read into it is still pending.

The kernels issue their fragment reads as inline-asm ds_read_b128 and wait for them with counted s_waitcnt lgkmcnt(N);
the compiler takes the asm's result for valid from the instruction on, and under register pressure it may copy such a
register (v_accvgpr_write, v_mov, a scratch store) before the data is there.  The scan is per basic block (pending reads
are forgotten at labels and branches, so a read that is still pending at a block boundary is not followed), LDS
operations return in order, and lgkmcnt(N) retires all but the N youngest.

usage: python tools/lds_hazard_check.py kernel.s      (exit status 1 if anything is flagged)
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update(m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add(m.group(1) + m.group(2))
    return out


def scan(path, limit=40):
    pending, flags = [], 0          # pending: (registers, line) of LDS operations in issue order
    for ln, line in enumerate(open(path).read().split("\n"), 1):
        t = line.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".") or t.endswith(":"):
            pending = []
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = []
            continue
        if op.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")):
            pending.append((regs(t.split()[1].rstrip(",")), ln))
            continue
        if op.startswith("ds_"):
            pending.append((set(), ln))
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            continue
        if op.startswith("s_"):
            continue
        used = regs(t)
        for rs, l0 in pending:
            if used & rs:
                flags += 1
                if flags <= limit:
                    print(f"line {ln}: {t[:80]}   <- LDS read of line {l0} pending: {sorted(used & rs)[:4]}")
                break
    return flags


if __name__ == "__main__":
    n = scan(sys.argv[1])
    print("flags:", n)
    sys.exit(1 if n else 0)
