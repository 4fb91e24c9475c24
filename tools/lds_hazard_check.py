"""Developer tool: scan a gfx950 assembly listing (hipcc -S --cuda-device-only) for instructions that touch a register
while an LDS read into it is still pending.

The kernels issue their fragment reads as inline-asm ds_read_b128 and wait for them with counted s_waitcnt lgkmcnt(N);
the compiler takes the asm's result for valid from the instruction on, and under register pressure it may copy such a
register (v_accvgpr_write, v_mov, a scratch store) before the data is there.  The scan is per basic block, LDS
operations return in order, and lgkmcnt(N) retires all but the N youngest.  Two more things are flagged, because the
counted waits rest on them:
  * a scalar memory load (s_load*, s_buffer_load*, s_scratch_load*, s_memtime ...) pending at a counted wait
    lgkmcnt(N > 0) that has LDS reads in front of it: scalar loads count on the same counter and return out of order, so
    "all but the N youngest" no longer says which reads have landed;
  * an inline-asm LDS read into registers still pending where a basic block ends (label, branch): it is not followed
    into the next block, so nothing would check its use there.
Both concern only reads the compiler does not see (between ;;#ASMSTART and ;;#ASMEND in the listing): for its own LDS
reads it places the waits itself, across blocks and with scalar loads in mind.

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


SMEM = ("s_load", "s_buffer_load", "s_scratch_load", "s_memtime", "s_memrealtime", "s_atc_probe", "s_dcache")


def scan(path, limit=40):
    pending, flags = [], 0          # pending: (registers, line) of LDS operations in issue order
    smem = 0                        # scalar memory operations pending on the same counter (they return out of order)

    def note(msg):
        nonlocal flags
        flags += 1
        if flags <= limit:
            print(msg)

    in_asm = False

    def block_ends(ln, t):
        nonlocal pending, smem
        for rs, l0, by_asm in pending:
            if rs and by_asm:
                note(f"line {ln}: {t[:60]}   <- block ends with the LDS read of line {l0} pending")
                break
        pending, smem = [], 0

    for ln, line in enumerate(open(path).read().split("\n"), 1):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.startswith(".") or t.endswith(":"):
            block_ends(ln, t)
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            block_ends(ln, t)
            continue
        if op.startswith(SMEM):
            smem += 1
            continue
        if op.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")):
            pending.append((regs(t.split()[1].rstrip(",")), ln, in_asm))
            continue
        if op.startswith("ds_"):
            pending.append((set(), ln, in_asm))
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending, smem = [], 0
                else:
                    if smem and any(rs and by_asm for rs, _, by_asm in pending):
                        note(f"line {ln}: {t[:60]}   <- counted wait with {smem} scalar memory load(s) pending on the same counter")
                    if n < len(pending):
                        pending = pending[len(pending) - n:]
            continue
        if op.startswith("s_"):
            continue
        used = regs(t)
        for rs, l0, _ in pending:
            if used & rs:
                note(f"line {ln}: {t[:80]}   <- LDS read of line {l0} pending: {sorted(used & rs)[:4]}")
                break
    return flags


if __name__ == "__main__":
    n = scan(sys.argv[1])
    print("flags:", n)
    sys.exit(1 if n else 0)
