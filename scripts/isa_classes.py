"""Instruction-class census of a sampling kernel's row loop and the issue-time bound it implies.

    python scripts/isa_classes.py [kernel-substring] [out-prefix]

Compiles tsim_amd/csrc/tsim_sample.hip to gfx950 assembly (device only), takes the named kernel (default: the fused
first pass of the benchmark shape, k_sample_lw_fast<2, 5>), finds its row loop (the longest backward branch) and
prices every instruction of the loop body with the issue cost measured on MI355X for its class
(profiles/r03/valu_table.txt, scripts/microbench/valu_table.hip; cycles per wave64 instruction and SIMD at the
clock the chip sustains for that class):

    full  2.4   v_add/sub/and/or/xor/mov/not/lshrrev/ashrrev/bitop3/fma/add_f32 on VGPR or constant operands
    half  4.3   everything else on the vector ALU: v_alignbit, v_lshlrev, v_add3, v_bcnt, v_cndmask, v_cmp, v_min/max,
                v_ffbl, v_mul, v_perm, ... and EVERY vector instruction with an SGPR operand
    salu  4.4   scalar ALU / scalar memory / branches (per SIMD; a separate issue port that overlaps the vector one
                across waves: scripts/microbench/salu_mix.hip)

Writes <out-prefix>.json (the census, the cycles per wave-row and the time per 10^6 shots they amount to on 1024
SIMDs) and <out-prefix>.s (the kernel's assembly) - the evidence behind bench.py's `valu.op_class_bound`.
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_bitop3_b32", "v_fma_f32", "v_add_f32", "v_sub_f32", "v_add_co_u32", "v_addc_co_u32"}
CYC = {"full": 2.4, "half": 4.3, "salu": 4.4}


def classify(line: str):
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_"):
        has_sgpr = bool(re.search(r"[ ,]s(\d+|\[\d+:\d+\])", line.split(None, 1)[1])) if len(line.split(None, 1)) > 1 else False
        if base in FULL and not has_sgpr and not op.endswith("sdwa"):
            return "full"
        return "half"
    if op in ("s_waitcnt", "s_nop", "s_endpgm", "s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_sample_lw_fastILi2ELi5E"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03", "first_pass_isa")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "tsim_amd", "csrc", "tsim_sample.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    meta = [l.strip("; \t") for l in lines[end:end + 90] if re.search(r"NumVgprs|TotalNumSgprs|Occupancy|ScratchSize|LDSByteSize", l)]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    lo, hi = max(loops, key=lambda ab: ab[1] - ab[0])
    # inner loops (the rank loop beyond the second ordinal, the direct-output runs) are counted once: one trip
    insts = [l.strip() for l in body[lo:hi + 1] if re.match(r"^\s*(v_|s_|global_|buffer_|flat_|ds_)", l)]
    cls = collections.Counter(classify(l) for l in insts)
    ops = collections.Counter(re.sub(r"_(e32|e64)$", "", l.split()[0]) for l in insts)
    valu_cycles = cls["full"] * CYC["full"] + cls["half"] * CYC["half"]
    salu_cycles = cls["salu"] * CYC["salu"]
    waves_per_simd = 1e6 / 64 / 1024  # wave-rows per SIMD for 10^6 shots
    res = {
        "kernel": want, "resources": meta, "loop_body_instructions": len(insts), "classes": dict(cls),
        "cycles_per_class": CYC, "valu_issue_cycles_per_wave_row": valu_cycles, "salu_issue_cycles_per_wave_row": salu_cycles,
        "top_ops": ops.most_common(24),
        "note": "static count of the row loop's body (inner loops once); issue cycles = sum over instructions of the measured "
                "cost of their class; per 10^6 shots = cycles x 15.26 wave-rows per SIMD / clock",
        "us_per_1e6_shots_at_2.3GHz": {"valu": valu_cycles * waves_per_simd / 2.3e3, "salu": salu_cycles * waves_per_simd / 2.3e3},
        "wave_instructions_per_1e6_shots": {"valu": (cls["full"] + cls["half"]) * 1e6 / 64, "valu_full_rate": cls["full"] * 1e6 / 64,
                                             "valu_half_rate": cls["half"] * 1e6 / 64, "scalar": cls["salu"] * 1e6 / 64},
    }
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out + ".json", "w"), indent=1)
    open(out + ".s", "w").write("\n".join(body) + "\n" + "\n".join("; " + m for m in meta) + "\n")
    print(json.dumps({k: res[k] for k in ("kernel", "classes", "valu_issue_cycles_per_wave_row", "salu_issue_cycles_per_wave_row",
                                          "us_per_1e6_shots_at_2.3GHz", "resources")}, indent=1))


if __name__ == "__main__":
    main()
