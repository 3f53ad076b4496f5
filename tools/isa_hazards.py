"""Static check of a hipcc -S listing: inline-asm MFMAs get no wait states from the compiler, so no VALU instruction may write a
register that an asm MFMA reads (A, B or C) within the two instructions before it (unless the asm string opens with s_nop), and
no compiler v_accvgpr_* / VALU may read an MFMA's D right behind it.  Usage: python tools/isa_hazards.py file.s [kernel-substring]

`--stores N file.s [kernel-substring]`: the 16-byte buffer-store pattern of the GEMM epilogues (dm_gemm_common.h, DM_EPI_BSTORE): a
`buffer_store_dwordx4` whose data registers are overwritten too soon after it stored the NEW value in some lanes on gfx950 (round 4; LLVM
pads nothing here when the store's soffset is a register).  For every such store the distance, in wait states (one per instruction, k + 1
per `s_nop k`), to the first later instruction that writes one of its data registers must be >= N; a label or branch ends the search."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'([va])(\d+)$', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def check(text, pick=""):
    bad = 0
    for f in re.split(r'\n(?=_Z\w+:)', text):
        name = f.split(':')[0]
        if pick not in name or 'v_mfma' not in f:
            continue
        ins = []          # (mnemonic, operands, in_asm, padded)
        in_asm, padded = False, False
        for l in f.split('\n'):
            t = l.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm, padded = True, False
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                continue
            parts = t.split(None, 1)
            mn, ops = parts[0], (parts[1] if len(parts) > 1 else '')
            if in_asm and mn == 's_nop':
                padded = True
            ins.append((mn, [o for o in ops.split(',')], in_asm, padded))
        n_mfma = 0
        for i, (mn, ops, ia, pad) in enumerate(ins):
            if not mn.startswith('v_mfma') or not ia:
                continue
            n_mfma += 1
            reads = set().union(*[regs(o) for o in ops[1:]])
            if not pad:
                for k in (1, 2):
                    if i - k < 0:
                        break
                    pm, pops, pia, _ = ins[i - k]
                    if pm.startswith('v_') and not pm.startswith('v_mfma') and pops and regs(pops[0]) & reads:
                        print(f"{name}: VALU write -> asm MFMA read without pad: {pm} {','.join(pops)}  ->  {mn} {','.join(ops)}")
                        bad += 1
            d = regs(ops[0])
            for k in (1, 2, 3):
                if i + k >= len(ins):
                    break
                nm, nops, nia, _ = ins[i + k]
                if nm.startswith('v_mfma'):
                    continue
                if nm.startswith('v_') and any(regs(o) & d for o in nops[1:]):
                    print(f"{name}: asm MFMA D read {k} instruction(s) later by {nm} {','.join(nops)}")
                    bad += 1
        print(f"{name}: {n_mfma} asm MFMAs checked")
    return bad


WRITERS = ('v_', 'buffer_load', 'global_load', 'ds_read', 'ds_bpermute', 'ds_permute', 'scratch_load', 'flat_load')


def check_stores(text, need, pick=""):
    """Minimum store -> data-register-overwrite distance per kernel; returns the number of stores closer than `need` wait states."""
    bad = 0
    for f in re.split(r'\n(?=_Z\w+:)', text):
        name = f.split(':')[0]
        if pick not in name or 'buffer_store_dwordx4' not in f:
            continue
        ins = []
        for l in f.split('\n'):
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            if t.endswith(':'):
                ins.append(('LABEL', []))
                continue
            parts = t.split(None, 1)
            ins.append((parts[0], [o for o in (parts[1] if len(parts) > 1 else '').split(',')]))
        n, worst, open_end = 0, None, 0
        for i, (mn, ops) in enumerate(ins):
            if not mn.startswith('buffer_store_dwordx4'):
                continue
            n += 1
            data = regs(ops[0])
            dist, found = 0, False
            for nm, nops in ins[i + 1:i + 200]:
                if nm == 'LABEL' or nm.startswith('s_cbranch') or nm.startswith('s_branch') or nm == 's_endpgm' or nm.startswith('s_setpc'):
                    break
                if nm.startswith(WRITERS) and not nm.startswith('v_cmp') and nops and regs(nops[0]) & data:
                    found = True
                    break
                dist += (int(nops[0], 0) + 1) if nm == 's_nop' else 1
            if not found:
                open_end += 1
                continue
            worst = dist if worst is None else min(worst, dist)
            if dist < need:
                print(f"{name}: {mn} {','.join(ops)}: data register overwritten {dist} wait states later (< {need})")
                bad += 1
        print(f"{name}: {n} 16-byte buffer stores, closest overwrite {worst} wait states later, {open_end} without one before the next branch / label")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "--stores":
        sys.exit(1 if check_stores(open(sys.argv[3]).read(), int(sys.argv[2]), sys.argv[4] if len(sys.argv) > 4 else "") else 0)
    sys.exit(1 if check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "") else 0)
