"""Build-time check of the generated code of conv_xk.hip (csrc/Makefile, run by __graft_entry__.build(); the kernel names of its two
predecessors, conv_xp / conv_xw, are still understood: the unit test feeds synthetic listings under them).

The kernel issues its matrix instructions as asm statements, so hipcc inserts none of the wait states an accumulator access needs
(MI355X: no hardware interlock between a matrix write and a vector read of the same register).  The sources are structured so that
hipcc never has a reason to touch an accumulator; this script proves it on the ISA of every instantiation:
  * the matrix instructions use exactly the expected accumulator tuples (conv_xp: 2 NT, conv_xw and conv_xk: 4 NT), the same registers throughout;
  * no v_accvgpr_mov / v_accvgpr_write (or any other non-matrix instruction) writes an accumulator register;
  * every read of an accumulator register sits behind the epilogue's tied wait (s_nop 15) IN THE SAME BASIC BLOCK with no matrix
    instruction in between: the "behind the wait" state is dropped at every label and after every branch, so a read reached through a
    back edge or a branch target from a block that ends in a matrix instruction cannot pass (round-4 advisor finding).
conv_xw keeps two operand fragment sets (64 registers) in the accumulator half of the register file as well: those are written by LDS
reads and read by matrix instructions only, and are not accumulators.
usage: check_xp_isa.py conv_xk.s | conv_xk.tune.s"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'a\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'a(\d+)', tok)
    return {int(m.group(1))} if m else set()


def check(name, lines):
    acc, tuples = set(), set()
    for ln in lines:
        if ln.startswith('v_mfma'):
            dst = ln.split()[1].rstrip(',')
            tuples.add(dst)
            acc |= regs(dst)
    m = re.search(r'conv_x([pwk])_kernelILi(\d)E', name)
    per_nt = 2 if m.group(1) == 'p' else 4
    nt = int(m.group(2))
    errs = []
    if len(tuples) != per_nt * nt or len(acc) != 16 * per_nt * nt:
        errs.append('%d accumulator tuples (%d registers), expected %d (%d)' % (len(tuples), len(acc), per_nt * nt, 16 * per_nt * nt))
    behind_tie = False
    for i, ln in enumerate(lines):
        if ln.endswith(':') or ln.startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
            behind_tie = False                       # a basic-block boundary: whatever protected the reads above does not reach across
            continue
        if ln.startswith('v_mfma'):
            behind_tie = False
            continue
        if ln.startswith('s_nop 15'):
            behind_tie = True
            continue
        toks = [t.rstrip(',') for t in ln.split()[1:]]
        touched = [t for t in toks if regs(t) & acc]
        if not touched:
            continue
        writes = bool(toks) and bool(regs(toks[0]) & acc) and not ln.startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store'))
        if writes:
            errs.append('line %d writes an accumulator: %s' % (i, ln))
        elif not behind_tie:
            errs.append('line %d reads an accumulator in the shadow of a matrix instruction: %s' % (i, ln))
    return nt, len(tuples), errs


def main(path):
    text = open(path).read().split('\n')
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r'^(_ZN3csd14conv_x[pwk]_kernel\w+):', ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            t = ln.strip()
            if t.startswith('s_endpgm'):
                kernels[name] = cur
                cur = None
            elif t and not t.startswith(';') and (not t.startswith('.') or t.split(';')[0].strip().endswith(':')):
                # directives (.p2align, .loc ...) are dropped; LABELS (.LBB0_3:) are kept - they are the basic-block boundaries check()
                # drops its "behind the wait" state at (round-5 advisor finding: they used to be filtered with the directives)
                cur.append(t.split(';')[0].strip())
    if not kernels:
        print('check_xp_isa: no conv_xk kernel in', path)
        return 1
    bad = 0
    for name, lines in sorted(kernels.items()):
        nt, ntup, errs = check(name, lines)
        print('%s: NT = %d, %d accumulator tuples, %d instructions: %s' % (name, nt, ntup, len(lines), 'ok' if not errs else 'FAILED'))
        for e in errs[:10]:
            print('   ', e)
        bad += bool(errs)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
