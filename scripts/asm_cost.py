"""Static VALU-issue cost of one kernel from its device assembly, basic block by basic block, with the per-class
issue costs measured by scripts/microbench/valu_rate.hip (cycles per wave64 instruction per SIMD on MI355X).
usage: python scripts/asm_cost.py file.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys

COST4 = ("v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_cmp", "v_max", "v_min", "v_cvt",
         "v_pk_", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_med3", "v_frexp", "v_ldexp", "v_rndne", "v_fract",
         "v_trunc", "v_floor", "v_ceil", "v_perm", "v_alignbit", "v_add_co", "v_addc", "v_sub_co", "v_subb", "v_add3", "v_and_or",
         "v_or3", "v_xad", "v_mbcnt", "v_add_lshl", "v_cmpx")
COST8 = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_exp_f32", "v_sin", "v_cos", "v_rcp_iflag")
COST16 = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")


def cost(op):
    if "_dpp" in op or "_sdwa" in op:
        return 4.2
    if op.startswith(COST16):
        return 16.3
    if op.startswith(COST8):
        return 8.2
    if "f64" in op or "b64" in op or "u64" in op or "i64" in op or op.startswith(COST4):
        return 4.3
    return 2.7


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read()
    names = [n for n in re.findall(r"^(_Z\S+):", txt, re.M) if key in n]
    name = names[0]
    body = txt[txt.index("\n" + name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    blocks, cur, label = [], collections.Counter(), "entry"
    tot = collections.Counter()
    for line in body.split("\n")[1:]:
        s = line.strip()
        if not s or s.startswith((";", ".")) and not s.startswith(".LBB"):
            continue
        if s.startswith(".LBB") and ":" in s:
            blocks.append((label, cur))
            label, cur = s.split(":")[0], collections.Counter()
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            c = cost(op)
            cur["valu"] += 1
            cur["cycles"] += c
            tot[re.sub(r"_e32|_e64", "", op)] += c
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
    blocks.append((label, cur))
    all_c = sum(b["cycles"] for _, b in blocks)
    print(f"{name[:90]}\nstatic: {sum(b['valu'] for _, b in blocks)} VALU, {all_c:.0f} issue cycles, "
          f"{sum(b['lds'] for _, b in blocks)} LDS, {sum(b['vmem'] for _, b in blocks)} VMEM, {sum(b['salu'] for _, b in blocks)} SALU")
    if "--blocks" in sys.argv:
        acc = 0.0
        for lab, b in blocks:
            acc += b["cycles"]
            if b["valu"] >= 8:
                print(f"  {lab:>12}: {b['valu']:4d} VALU {b['cycles']:7.0f} cyc  (cum {acc:7.0f})  lds {b['lds']:3d} vmem {b['vmem']:3d}")
    print("top opcodes by issue cycles:")
    for op, c in tot.most_common(22):
        print(f"  {op:24s} {c:7.0f}  {100 * c / all_c:4.1f} %")


main()
