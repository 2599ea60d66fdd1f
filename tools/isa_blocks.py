"""Per-basic-block instruction census of one kernel (which blocks hold the SGPR-spill traffic, v_readlane / v_writelane):

    hipcc <flags of csrc/Makefile> -gline-tables-only -x hip packet.hip --cuda-device-only -S -o /tmp/packet_g.s
    python tools/isa_blocks.py /tmp/packet_g.s k_packetILi2 [--all]

Per block: vector ALU instructions, v_readlane (rl) / v_writelane (wl), scalar ALU, scalar memory, LDS, vector memory, branch targets and the
three source lines most instructions of the block come from (file:line as in the .file table).  Static counts; weigh with tools/pk_counters.py."""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    on, cur, blocks, b = False, None, [], None
    new = lambda label: {'label': label, 'valu': 0, 'rl': 0, 'wl': 0, 'salu': 0, 'smem': 0, 'br': 0, 'ds': 0, 'vmem': 0, 'lines': collections.Counter(), 'targets': []}
    for line in open(path):
        if re.match(r'^_Z\w*%s\w*:' % kern, line):
            on = True; b = new('entry'); blocks.append(b); continue
        if not on:
            continue
        if 's_endpgm' in line:
            break
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
        if m:
            cur = (int(m.group(1)), int(m.group(2))); continue
        m = re.match(r'^(\.LBB\d+_\d+):', line)
        if m:
            b = new(m.group(1)); blocks.append(b); continue
        t = line.strip().split()
        if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
            continue
        op = t[0]
        if op == 'v_readlane_b32': b['rl'] += 1
        elif op == 'v_writelane_b32': b['wl'] += 1
        elif op.startswith('v_'): b['valu'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): b['br'] += 1; b['targets'].append(t[1])
        elif op.startswith('s_load'): b['smem'] += 1
        elif op.startswith(('s_waitcnt', 's_nop')): pass
        elif op.startswith('s_'): b['salu'] += 1
        elif op.startswith('ds_'): b['ds'] += 1
        else: b['vmem'] += 1
        b['lines'][cur] += 1
    tot = collections.Counter()
    for i, b in enumerate(blocks):
        for k in ('valu', 'rl', 'wl', 'salu', 'smem', 'ds', 'vmem'):
            tot[k] += b[k]
        if '--all' in sys.argv or b['rl'] + b['wl'] > 0 or b['valu'] >= 20:
            ls = ','.join('%d:%d' % k for k, _ in b['lines'].most_common(3) if k)
            print('%3d %-12s valu %3d rl %2d wl %2d salu %3d smem %2d ds %2d vm %2d -> %s  [%s]' % (i, b['label'], b['valu'], b['rl'], b['wl'], b['salu'], b['smem'], b['ds'], b['vmem'], ' '.join(b['targets']), ls))
    print('total', dict(tot))


main()
