"""Static VALU issue cost of the blend kernels' visit loops: instruction counts of the inner loop (from the compiler's ISA)
weighted with the issue cycles measured by tools/valu_rate.hip on gfx950 (profiles/r02_a_valu_rate.json, 8 waves per SIMD).

  python tools/isa_cost.py            (needs hipcc; prints a JSON summary, written to profiles/r02_c_blend_issue_cost.json by the
                                       round's GPU session)

VERDICT r01 item 3: the issue rate is NOT one number -- plain fp32 ops cost ~2.5-3 cycles per wave64 instruction, DPP / packed /
select / compare / integer-mad forms ~4.2-5.6, transcendental and v_permlane*_swap ~8.2 -- so a utilisation figure needs the
instruction mix."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "photo-slam_amd", "csrc")
RATE = os.path.join(ROOT, "profiles", "r02_a_valu_rate.json")

# mnemonic prefix -> microbenchmark entry
CLASS = [
    ("v_exp_f32", "v_exp_f32"), ("v_rcp_f32", "v_rcp_f32"), ("v_log_f32", "v_exp_f32"), ("v_sqrt_f32", "v_rcp_f32"), ("v_rsq_f32", "v_rcp_f32"),
    ("v_permlane32_swap", "v_permlane32_swap"), ("v_permlane16_swap", "v_permlane16_swap"),
    ("v_pk_fma_f32", "v_pk_fma_f32"), ("v_pk_mul_f32", "v_pk_mul_f32"), ("v_pk_add_f32", "v_pk_add_f32"),
    ("v_cndmask_b32", "v_cndmask_b32_sgpr_mask"), ("v_cmp", "v_cmp_gt_f32_e32_vcc"),
    ("v_fma_f32", "v_fma_f32"), ("v_fmac_f32", "v_fmac_f32_e32"), ("v_mul_f32", "v_mul_f32"), ("v_add_f32", "v_add_f32"),
    ("v_sub_f32", "v_sub_f32"), ("v_min_f32", "v_min_f32"), ("v_max_f32", "v_min_f32"), ("v_mad_u32_u24", "v_mad_u32_u24"),
    ("v_lshl_add_u32", "v_lshl_add_u32"), ("v_mov_b32", "v_mov_b32_from_sgpr"), ("v_accvgpr", "v_mov_b32_from_sgpr"),
]
DEFAULT = "v_lshl_add_u32"   # other integer / bit ops: the 4-cycle class


def cost_table():
    d = json.load(open(RATE))["ops"]
    return {k: v["w8"]["event_cycles"] for k, v in d.items()}


def classify(line, table):
    m = line.split()[0]
    if "dpp" in line or m.endswith("_dpp"):
        return "dpp", table["v_add_f32_dpp_row_shr"]
    for prefix, key in CLASS:
        if m.startswith(prefix):
            return prefix, table[key]
    return "other:" + m, table[DEFAULT]


def visit_loop(asm, kernel):
    """instructions of the innermost loop that contains v_exp_f32 (the per (quad, entry) visit) of `kernel`"""
    start = asm.index(kernel + ":")
    end = asm.index(".Lfunc_end", start)
    lines = asm[start:end].splitlines()
    iexp = next(i for i, l in enumerate(lines) if "v_exp_f32" in l)
    # loop header: the closest preceding label with "Inner Loop Header"
    ihead = max(i for i in range(iexp) if re.match(r"^\.LBB\d+_\d+:", lines[i]) and "Inner Loop Header" in "".join(lines[i:i + 5]))
    label = lines[ihead].split(":")[0]
    # the loop ends at the last branch back to the header after iexp
    backs = [i for i in range(iexp, len(lines)) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b|s_branch\s+" + re.escape(label) + r"\b", lines[i])]
    iend = backs[-1] if backs else len(lines) - 1
    # follow the blocks between header and back edge (straight-line approximation: every block in between)
    body = [l.strip() for l in lines[ihead:iend + 1] if l.strip() and not l.strip().startswith((";", "."))]
    for i, l in enumerate(body):   # the segment write-out behind the workgroup barrier is not part of a visit
        if l.startswith("s_barrier"):
            body = body[:i]
            break
    return body


def analyse(src, kernel, flags):
    out = subprocess.check_output(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                                   "-fno-gpu-rdc", "-fno-slp-vectorize", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", "-"] + flags,
                                  stderr=subprocess.DEVNULL, text=True)
    name = next(l.split(":")[0] for l in out.splitlines() if re.match(r"^_Z\w*" + kernel + r"\w*:", l))
    body = visit_loop(out, name)
    table = cost_table()
    valu = [l for l in body if l.startswith("v_")]
    by = {}
    total = 0.0
    for l in valu:
        k, c = classify(l, table)
        n, t = by.get(k, (0, 0.0))
        by[k] = (n + 1, t + c)
        total += c
    lds = sum(1 for l in body if l.startswith("ds_"))
    salu = sum(1 for l in body if l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop")))
    return dict(kernel=kernel, valu_instructions=len(valu), valu_issue_cycles=round(total, 1), lds_instructions=lds, salu_instructions=salu,
                mean_cycles_per_valu=round(total / max(len(valu), 1), 2),
                classes={k: dict(n=n, cycles=round(t, 1)) for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])})


if __name__ == "__main__":
    res = {"cost_source": "profiles/r02_a_valu_rate.json (event_cycles at 8 waves per SIMD)",
           "note": "straight-line count of the visit loop body: header to back edge, the rarely taken blocks included",
           "blend_fwd": analyse("blend_fwd.hip", "blend_fwd_kernel", []),
           "blend_bwd": analyse("blend_bwd.hip", "blend_bwd_kernel", [])}
    json.dump(res, sys.stdout, indent=1)
    print()
