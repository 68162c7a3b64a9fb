"""dev: instruction mix of one kernel in the ISA listing tools/kcheck.sh leaves in /tmp/lpt_kernels.s
usage: python tools/dev/isa_count.py <mangled-name substring> [first-label last-label]   (e.g. k_traceILb0ELb1ELb0)"""
import collections
import re
import sys

lines = open("/tmp/lpt_kernels.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4lptd\w*%s\w*:" % re.escape(sys.argv[1]), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = [l.strip() for l in lines[start + 1:end]]
body = [l for l in body if l and not l.startswith(";") and not l.startswith(".") or re.match(r"^\.LBB\d+_\d+:", l)]
if len(sys.argv) > 3:
    a = next(i for i, l in enumerate(body) if l.startswith(sys.argv[2] + ":"))
    b = next(i for i, l in enumerate(body) if l.startswith(sys.argv[3] + ":"))
    body = body[a:b]
ins = [l.split()[0] for l in body if not l.endswith(":")]
c = collections.Counter(ins)
fast = {"v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_mov_b32_e32", "v_not_b32_e32", "v_lshrrev_b32_e32",
        "v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_fma_f32", "v_fmac_f32_e32", "v_mul_f32_e64", "v_add_f32_e64", "v_sub_f32_e64", "v_and_b32_e64", "v_or_b32_e64"}
v = sum(n for k, n in c.items() if k.startswith("v_"))
vf = sum(n for k, n in c.items() if k in fast)
print("%d instructions: valu %d (fast class %d), salu %d, vmem %d, lds %d, waitcnt %d" % (
    len(ins), v, vf, sum(n for k, n in c.items() if k.startswith("s_") and not k.startswith("s_waitcnt")),
    sum(n for k, n in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))), sum(n for k, n in c.items() if k.startswith("ds_")), c.get("s_waitcnt", 0)))
for k, n in c.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 40):
    print("%5d %s" % (n, k))
