"""Static check of the hand-ordered loads of conv2d_ws_pp_kernel (kernels_conv2d_ws.hip).

The kernel issues its global loads from `asm volatile` statements the compiler's
waitcnt pass does not see and waits for them with ONE explicit `s_waitcnt vmcnt(0)`;
between a load and that wait nothing may read or write the destination registers
(a register-allocator copy or a spill there would move a value that has not arrived).
This walks the control-flow graph of the compiler's assembly (forward data flow to a
fixed point, union over predecessors): from an inline-asm `global_load_dwordx4 v[a:b]`
to the next inline-asm `s_waitcnt vmcnt(0)` on every path no instruction may mention
v[a:b].  Also: no scratch.

Usage: python tools/check_inflight.py [asm file]  (default: compiles the kernel file)
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, '..', 'sup3r_amd', 'csrc', 'kernels_conv2d_ws.hip')


def compile_asm():
    out = os.path.join(tempfile.mkdtemp(), 'ws.s')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                    os.path.abspath(SRC), '-o', out], check=True, cwd=os.path.dirname(os.path.abspath(SRC)),
                   stderr=subprocess.DEVNULL)
    return out


def regs_of(tok):
    """v[a:b] or vN -> set of register numbers"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def mentioned(line):
    s = set()
    for tok in re.findall(r'v\[\d+:\d+\]|\bv\d+\b', line):
        s |= regs_of(tok)
    return s


def parse_blocks(body):
    """body: list of (line number, text) of one function -> basic blocks and their successors"""
    blocks, cur, labels = [], {'label': None, 'ins': []}, {}
    in_asm = False
    for ln, raw in body:
        t = raw.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';'):
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            if cur['ins'] or cur['label']:
                blocks.append(cur)
            cur = {'label': m.group(1), 'ins': []}
            continue
        if t.startswith('.') or t.endswith(':'):
            continue
        code = t.split(';')[0].strip()
        if not code:
            continue
        cur['ins'].append((ln, code, in_asm))
        op = code.split()[0]
        if op in ('s_branch', 's_endpgm') or op.startswith('s_cbranch'):
            blocks.append(cur)
            cur = {'label': None, 'ins': []}
    if cur['ins'] or cur['label']:
        blocks.append(cur)
    for k, b in enumerate(blocks):
        if b['label']:
            labels[b['label']] = k
    for k, b in enumerate(blocks):
        succ = []
        last = b['ins'][-1][1] if b['ins'] else ''
        op = last.split()[0] if last else ''
        if op == 's_endpgm':
            pass
        elif op == 's_branch':
            succ.append(labels[last.split()[1]])
        else:
            if op.startswith('s_cbranch'):
                succ.append(labels[last.split()[1]])
            if k + 1 < len(blocks):
                succ.append(k + 1)
        b['succ'] = succ
    return blocks


def transfer(b, live_in, problems, name):
    """walk one block; live: register -> line of the load in flight"""
    live = dict(live_in)
    for ln, code, in_asm in b['ins']:
        if 'scratch_' in code and problems is not None:
            problems.append('%s: scratch access at line %d: %s' % (name, ln, code))
        if in_asm and code.startswith('global_load_dwordx4'):
            dst = regs_of(code.split()[1].rstrip(','))
            if problems is not None:
                for r in (mentioned(code) - dst) & set(live):
                    problems.append('%s: line %d reads v%d (in flight since line %d): %s'
                                    % (name, ln, r, live[r], code))
            for r in dst:
                live[r] = ln
        elif in_asm and code.startswith('s_waitcnt') and 'vmcnt(0)' in code:
            live = {}
        elif live:
            hit = mentioned(code) & set(live)
            if problems is not None:
                for r in hit:
                    problems.append('%s: line %d touches v%d (in flight since line %d): %s'
                                    % (name, ln, r, live[r], code))
    return live


def check(path):
    lines = open(path).read().split('\n')
    problems, n_loads, n_kern = [], 0, 0
    i = 0
    while i < len(lines):
        if re.match(r'^_ZN.*conv2d_ws_pp_kernel.*:\s', lines[i] + ' '):
            n_kern += 1
            name = lines[i].split(':')[0][-40:]
            j = i + 1
            body = []
            while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
                body.append((j + 1, lines[j]))
                j += 1
            blocks = parse_blocks(body)
            n_loads += sum(1 for b in blocks for _, c, a in b['ins'] if a and c.startswith('global_load_dwordx4'))
            # forward dataflow to a fixed point (union over predecessors)
            live_in = [dict() for _ in blocks]
            work = [0]
            seen_out = [None] * len(blocks)
            while work:
                k = work.pop()
                out = transfer(blocks[k], live_in[k], None, name)
                if seen_out[k] is not None and set(out) <= set(seen_out[k]):
                    continue
                so = dict(seen_out[k] or {}); so.update(out); seen_out[k] = so
                for sidx in blocks[k]['succ']:
                    merged = dict(live_in[sidx]); merged.update(out)
                    if set(merged) != set(live_in[sidx]) or seen_out[sidx] is None:
                        live_in[sidx] = merged
                        work.append(sidx)
            for k, b in enumerate(blocks):
                transfer(b, live_in[k], problems, name)
            i = j
        i += 1
    return n_kern, n_loads, problems


if __name__ == '__main__':
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    nk, nl, pr = check(path)
    print('%d kernels, %d hand-ordered loads, %d problems' % (nk, nl, len(pr)))
    for p in pr[:40]:
        print('  ' + p)
    sys.exit(1 if pr or nk == 0 else 0)
