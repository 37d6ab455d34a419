"""Static instruction census of one kernel in a device assembly file (development aid).

usage: census.py file.s kernel_name_substring [--blocks]
Counts VALU / SALU / LDS / VMEM / branch instructions per basic block and for the largest loop (the span between the
backward branch with the longest reach and its target); --blocks prints every block with its label and successors.
The file comes from: hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o file.s source.hip
"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "BR"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    path, name = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0] and not l.startswith(".L"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start + 1:end]
    # instruction list with block labels
    insts = []          # (index, op, text, label_before)
    labels = {}
    pending = []
    for l in body:
        s = l.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            pending.append(m.group(1))
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        for p in pending:
            labels[p] = len(insts)
        insts.append((op, s, list(pending)))
        pending = []
    total = {}
    for op, s, _ in insts:
        c = classify(op)
        total[c] = total.get(c, 0) + 1
    print("kernel %s: %d instructions  %s" % (lines[start].split(":")[0][:60], len(insts), total))
    # backward branches
    loops = []
    for i, (op, s, _) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((i - labels[tgt], labels[tgt], i, tgt))
    loops.sort(reverse=True)
    for span, a, b, tgt in loops[:12]:
        cnt = {}
        for op, s, _ in insts[a:b + 1]:
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
        print("  loop %-12s insts %6d..%6d (%5d): %s" % (tgt, a, b, span + 1, " ".join("%s=%d" % kv for kv in sorted(cnt.items()))))
    if show_blocks:
        cur = None
        cnt = {}
        for i, (op, s, lab) in enumerate(insts):
            if lab:
                if cur is not None:
                    print("   ", cur, cnt)
                cur, cnt = "%s@%d" % (",".join(lab), i), {}
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
            if c == "BR":
                cnt.setdefault("to", []).append(s.split()[-1])
        print("   ", cur, cnt)
    # op histogram of the largest loop
    if loops and "--hist" in sys.argv:
        span, a, b, tgt = loops[0]
        h = {}
        for op, s, _ in insts[a:b + 1]:
            h[op] = h.get(op, 0) + 1
        for op, n in sorted(h.items(), key=lambda kv: -kv[1])[:60]:
            print("    %-28s %d" % (op, n))


if __name__ == "__main__":
    main()
