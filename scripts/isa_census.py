#!/usr/bin/env python3
"""ISA census of the decode kernels' byte loops: opcode x issue class, static, from hipcc -S output, next to
the EXECUTED instruction counts per byte and wave of the rocprofv3 SQ_INSTS_* counters (profiles/<tag>_<config>_summary.txt).

    python scripts/isa_census.py <tag>      (compiles lit_kernels.hip and lit_decode2.hip to /tmp, reads profiles/<tag>_*_summary.txt)

Issue classes (scripts/ubench/issue_rates.hip, profiles/r03a_ubench_issue_rates.txt, cycles per wave64 instruction and SIMD at 8
waves per SIMD): A ~2.4 (mov, and/or/xor/not, add/sub_u32, lshrrev_b32, ashrrev, bitop3, add/mul/fma f32), B ~4.2 (every other
VALU op incl. the 64-bit ones, DPP, cmp, cndmask, cvt, mul24/mad24, lshl, ffbl), R ~8.2 (v_rcp_f32).
The byte loop of a kernel = the smallest loop that holds all of its row loads (buffer_load_ushort)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS_A = {"v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
           "v_bitop3_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_sub_f32"}
COST = {"A": 2.4, "B": 4.2, "R": 8.2}


def issue_class(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_rcp"):
        return "R"
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "B"
    return "A" if base in CLASS_A else "B"


def kernel_body(asm, pattern):
    m = re.search(r"^(_Z\w*%s\w*):.*\n" % pattern, asm, re.M)
    if not m:
        return None, None
    body = asm[m.end():asm.index(".Lfunc_end", m.end())]
    ins, labels = [], {}
    for l in body.split("\n"):
        l = l.split(";")[0].strip()
        if not l or l.startswith("//"):
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(ins); continue
        if l.startswith("."):
            continue
        ins.append(l)
    return ins, labels


def byte_loop(ins, labels):
    loops = []
    for i, x in enumerate(ins):
        mm = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", x)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in labels and labels[t] <= i:
                loops.append((labels[t], i))
    # nest: stream loop (holds the table initialisation, buffer_store_dwordx4) > chunk loop > 16-byte window loop > byte loop.
    # The byte loop is the smallest loop that holds every row load (buffer_load_ushort) of the chunk loop.
    nrow = lambda a, b: sum("buffer_load_ushort" in y for y in ins[a:b + 1])
    inner = [(a, b) for a, b in loops if nrow(a, b) and not any("buffer_store_dwordx4" in y for y in ins[a:b + 1])]
    most = max(nrow(a, b) for a, b in inner)
    return min(((a, b) for a, b in inner if nrow(a, b) == most), key=lambda z: z[1] - z[0])


def census(ins):
    ops = collections.Counter(x.split()[0] for x in ins)
    rows, tot = [], collections.Counter()
    for op, n in ops.items():
        if op.startswith("v_"):
            c = issue_class(op); tot[c] += n; rows.append((c, op, n))
        elif op.startswith("s_"):
            tot["salu"] += n
        elif op.startswith("ds_"):
            tot["lds"] += n
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            tot["vmem"] += n
    return rows, tot


def executed(tag, config):
    p = os.path.join(ROOT, "profiles", f"{tag}_{config}_summary.txt")
    vals = {}
    if os.path.exists(p):
        for line in open(p):
            m = re.match(r".*lit_decode.*?\s(SQ_\w+|GRBM_GUI_ACTIVE)\s+avg=(\S+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
    return vals


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03b"
    out = []
    tmp = "/tmp/isa_census"
    os.makedirs(tmp, exist_ok=True)
    for src in ("lit_kernels.hip", "lit_decode2.hip"):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", os.path.join(tmp, src + ".s"),
                        os.path.join(ROOT, "divans_amd", "csrc", src)], check=True, capture_output=True)
    kernels = [
        ("generation 1, configs[1] (TestSimple): lit_decode_kernel<4,true,false,2,false>", "lit_kernels.hip", "lit_decode_kernelILi4ELb1ELb0ELi2ELb0E", "simple", 65536),
        ("generation 3, configs[1] (TestSimple): lit_decode2_kernel_w7<4,true,false,false,17>", "lit_decode2.hip", "lit_decode2_kernel_w7ILi4ELb1ELb0ELb0ELi17E", "simple", 65536),
        ("generation 1, configs[2]/[3] (TestContextMixing): lit_decode_kernel<4,false,true,2,false>", "lit_kernels.hip", "lit_decode_kernelILi4ELb0ELb1ELi2ELb0E", "mixing", 65536),
        ("generation 3, configs[2]/[3] (TestContextMixing): lit_decode2_kernel_w7<4,false,true,false,19>", "lit_decode2.hip", "lit_decode2_kernel_w7ILi4ELb0ELb1ELb0ELi19E", "mixing", 65536),
    ]
    for title, src, pat, config, streams in kernels:
        asm = open(os.path.join(tmp, src + ".s")).read()
        ins, labels = kernel_body(asm, pat)
        if ins is None:
            out.append(f"== {title}: not found"); continue
        a, b = byte_loop(ins, labels)
        rows, tot = census(ins[a:b + 1])
        valu = tot["A"] + tot["B"] + tot["R"]
        cyc = sum(COST[c] * tot[c] for c in "ABR")
        out.append(f"== {title}")
        out.append(f"   byte loop: {b - a + 1} instructions static = {valu} VALU (class A {tot['A']}, B {tot['B']}, R {tot['R']}; {cyc:.0f} issue cycles if all executed), "
                   f"{tot['salu']} SALU, {tot['lds']} LDS, {tot['vmem']} VMEM")
        out.append("   (static = every path of the loop once: refill, cache-miss, renormalisation and absent-cache paths included)")
        for c in "ABR":
            line = ", ".join(f"{op} {n}" for cc, op, n in sorted(rows, key=lambda r: (-r[2], r[1])) if cc == c)
            out.append(f"   class {c}: {line}")
        if src == "lit_decode2.hip" or "generation 1" in title:
            ex = executed(tag if "generation 3" in title else "r02c", config)
            if ex:
                bw = streams * 65536 / 4.0
                out.append(f"   executed ({'profiles/' + (tag if 'generation 3' in title else 'r02c') + '_' + config + '_summary.txt'}, per byte and wave = per 4 stream-bytes): "
                           + ", ".join(f"{k[3:] if k.startswith('SQ_') else k} {v / bw:.1f}" for k, v in sorted(ex.items()) if k.startswith("SQ_INSTS")))
                if "SQ_ACTIVE_INST_VALU" in ex and "GRBM_GUI_ACTIVE" in ex:
                    avail = ex["GRBM_GUI_ACTIVE"] / 8.0 * 1024 / 4.0      # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_* count quad-cycles
                    out.append(f"   VALU busy: SQ_ACTIVE_INST_VALU {ex['SQ_ACTIVE_INST_VALU']:.4g} quad-cycles of {avail:.4g} available on 1024 SIMDs = {100 * ex['SQ_ACTIVE_INST_VALU'] / avail:.0f} %"
                               + (f"; waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) {100 * ex['SQ_WAIT_ANY'] / ex['SQ_WAVE_CYCLES']:.0f} %" if "SQ_WAIT_ANY" in ex and "SQ_WAVE_CYCLES" in ex else ""))
        out.append("")
    text = "\n".join(out)
    open(os.path.join(ROOT, "profiles", f"{tag}_isa_census.txt"), "w").write(__doc__.split("\n\n")[0] + "\n\n" + text)
    print(text)


if __name__ == "__main__":
    main()
