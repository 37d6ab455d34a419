"""Weighted static census of an instruction range of a kernel (development aid): pipe cycles by the issue costs measured
with tools/micro/valu_rate3.hip.  usage: census_cost.py file.s kernel first last [--hist]"""
import re, sys
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32_e", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_cndmask",
        "v_cmp", "v_bitop3", "v_not_b32", "v_addc", "v_add_co", "v_sub_co", "v_subb")
def cost(op):
    if op.startswith("v_qsad"): return 17.0
    if op.startswith(FAST): return 2.6
    if op.startswith("v_"): return 4.5
    return 0.0
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0]][0]
end = start
while not lines[end].startswith(".Lfunc_end"): end += 1
ins = []
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith((";", "//", ".")) or re.match(r"^\.?LBB", s): continue
    ins.append(s.split()[0])
a, b = int(sys.argv[3]), int(sys.argv[4])
seg = ins[a:b]
tot = sum(cost(o) for o in seg)
cl = {}
for o in seg:
    k = "VALU" if o.startswith("v_") else "LDS" if o.startswith("ds_") else "VMEM" if o.startswith(("global_", "scratch_", "buffer_")) else "SALU/other"
    cl[k] = cl.get(k, 0) + 1
print("instructions %d..%d: %s  weighted VALU pipe cycles %.0f" % (a, b, cl, tot))
if "--hist" in sys.argv:
    h = {}
    for o in seg: h[o] = h.get(o, 0) + 1
    for o, n in sorted(h.items(), key=lambda kv: -kv[1] * max(cost(kv[0]), 1))[:40]:
        print("   %-28s %4d  x %.1f = %.0f" % (o, n, cost(o), n * cost(o)))
