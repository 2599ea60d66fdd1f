"""Static instruction census of one kernel by SOURCE LINE (which source lines the scalar / vector instructions of a kernel come from).

    hipcc <flags of csrc/Makefile> -gline-tables-only -x hip packet.hip --cuda-device-only -S -o /tmp/packet_g.s
    python tools/isa_by_line.py /tmp/packet_g.s k_packetILi2 [--top 40]

Instructions are attributed to the last `.loc file line` before them (inlined code is attributed to the inlined function's own line).
Static counts, not executed counts: multiply by the walk counters of tools/pk_counters.py to weigh them."""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    files, cur, on = {}, None, False
    cnt = collections.defaultdict(lambda: [0, 0, 0, 0])   # salu, branch, smem, valu+other
    for line in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if re.match(r'^_Z\w*%s\w*:' % kern, line):
            on = True
            continue
        if not on:
            continue
        if "s_endpgm" in line:
            break
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith(("s_cbranch", "s_branch")):
            k = 1
        elif op.startswith(("s_load", "s_buffer_load")):
            k = 2
        elif op.startswith(("s_waitcnt", "s_nop")):
            continue
        elif op.startswith("s_"):
            k = 0
        else:
            k = 3
        cnt[cur][k] += 1
    rows = sorted(cnt.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))
    tot = [sum(v[i] for v in cnt.values()) for i in range(4)]
    print("total: salu %d branch %d smem %d vector/other %d" % tuple(tot))
    for (f, l), v in rows[:top]:
        print("%-16s %5d  salu %4d branch %3d smem %2d vec %4d" % (files.get(f, "?"), l, v[0], v[1], v[2], v[3]))


if __name__ == "__main__":
    main()
