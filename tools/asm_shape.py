"""Developer tool: one letter per instruction of a kernel's basic blocks (M mfma, e exp, c cvt_pk, d ds_read,
G global/buffer load, S scratch, w waitcnt, B barrier, j branch, v other vector, . scalar).
usage: asm_shape.py <file.s> <kernel name fragment>"""
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index(':\n', s.index(sys.argv[2], i))
j = s.index('.Lfunc_end', i)
out = []
for l in s[i:j].split('\n'):
    l = l.strip()
    if not l or l.startswith(';'):
        continue
    if l.startswith('.LBB'):
        out.append('\n' + l.split(':')[0] + ': ')
        continue
    op = l.split()[0]
    for pre, ch in (('v_mfma', 'M'), ('v_exp', 'e'), ('v_cvt_pk', 'c'), ('ds_read', 'd'), ('ds_', 'D'), ('buffer_load', 'G'),
                    ('global_load', 'G'), ('scratch_', 'S'), ('s_waitcnt', 'w'), ('s_barrier', 'B'), ('s_cbranch', 'j'),
                    ('s_branch', 'j'), ('s_nop', 'n'), ('v_', 'v')):
        if op.startswith(pre):
            out.append(ch)
            break
    else:
        out.append('.')
print(''.join(out))
